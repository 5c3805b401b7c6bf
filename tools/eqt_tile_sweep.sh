# On-box sweep of the wave / tile shape of the EQT decoder stages 3-5 (conv_mfma ConvCfg WM, WN, NW): rebuilds the
# library per variant and prints the per-kernel times of bench.py.
set -e
cd $GRAFT_REPO_ROOT
F=volpick_amd/csrc/eqt.hip
run() { # name  d3(WM WN NW)  d4(WM WN NW)  d5(WM WN NW)
  sed -e "s/using EQ_d3 = ConvCfg<32, 0, 32, 2, 5, 1, -2, 0, [0-9]*, [0-9]*, [0-9]*, 1, EPI_STORE>;/using EQ_d3 = ConvCfg<32, 0, 32, 2, 5, 1, -2, 0, $2, $3, $4, 1, EPI_STORE>;/" \
      -e "s/using EQ_d4 = ConvCfg<32, 0, 16, 2, 5, 1, -2, 0, [0-9]*, [0-9]*, [0-9]*, 1, EPI_STORE>;/using EQ_d4 = ConvCfg<32, 0, 16, 2, 5, 1, -2, 0, $5, $6, $7, 1, EPI_STORE>;/" \
      -e "s/using EQ_d5 = ConvCfg<16, 0, 16, 2, 5, 1, -2, 0, [0-9]*, [0-9]*, [0-9]*, 1, EPI_STORE>;/using EQ_d5 = ConvCfg<16, 0, 16, 2, 5, 1, -2, 0, $8, $9, ${10}, 1, EPI_STORE>;/" \
      -i $F
  make -C volpick_amd/csrc -j8 > /dev/null 2>&1
  echo "== $1: d3 ($2 $3 $4)  d4 ($5 $6 $7)  d5 ($8 $9 ${10})"
  python bench.py --model eqtransformer --no-cpu-baseline --steps 60 --warmup 6 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), [(k['name'][:9], round(k['ms']*1e3,1)) for k in d['forward']['kernels'] if k['name'] in ('decoder.3','decoder.4','decoder.5')])"
}
run base   2 2 6  2 2 6  2 2 6
run a      4 1 6  1 4 6  1 4 6
run b      1 4 3  1 4 3  1 4 3
run c      2 2 4  2 2 4  2 2 4
run d      4 1 3  2 2 8  2 2 8
run base   2 2 6  2 2 6  2 2 6
