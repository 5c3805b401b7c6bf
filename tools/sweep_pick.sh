# On-box sweep of the workgroup size of the fused EQT middle kernel (and, earlier, of its separate predecessors): a big
# workgroup that mostly waits (47 sequential LSTM steps of ONE wave, four times over) holds wave slots and registers the
# other device contexts' MFMA kernels could use.
set -e
cd $GRAFT_REPO_ROOT
for n in 1024 512 256 128; do
  sed -i "s/constexpr int MID_NTH = [0-9]*;/constexpr int MID_NTH = $n;/" volpick_amd/csrc/eqt_kernels.hip
  make -C volpick_amd/csrc -j8 > /dev/null 2>&1
  echo "== fused middle kernel: $n threads"
  for i in 1 2; do python bench.py --model eqtransformer --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], [(k['name'][:9], round(k['ms']*1e3,1)) for k in d['forward']['kernels'] if 'mid' in k['name']])"; done
done
