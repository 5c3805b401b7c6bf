# On-box sweep of the workgroup size of the two latency-bound EQT kernels that used 1024 threads: a big workgroup that mostly
# waits (47 sequential LSTM steps of ONE wave in pick_branch) holds wave slots the other device contexts' kernels could use.
set -e
cd $GRAFT_REPO_ROOT
for n in 1024 512 256; do
  sed -i "s/constexpr int TR_NTH = [0-9]*;/constexpr int TR_NTH = $n;/" volpick_amd/csrc/eqt_kernels.hip
  make -C volpick_amd/csrc -j8 > /dev/null 2>&1
  echo "== transformer threads $n (pick_branch $(grep -o 'PICK_NTH = [0-9]*' volpick_amd/csrc/eqt_kernels.hip | head -1))"
  for i in 1 2; do python bench.py --model eqtransformer --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], [(k['name'][:8], round(k['ms']*1e3,1)) for k in d['forward']['kernels'] if 'pick' in k['name'] or 'transf' in k['name']])"; done
done
