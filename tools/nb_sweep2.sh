# On-box sweep of the n-tiles-per-item of the weight-heavy core layers (the first sweep, tools/nb_sweep.sh, covered the
# three 16/32-channel "same" layers): fewer n-tiles per item = more waves streaming weights concurrently.
set -e
cd $GRAFT_REPO_ROOT
F=volpick_amd/csrc/phasenet_fused.hip
run() { # name u0same u1same d3same d2same
  sed -e "s/using C_u0same = LdsLayer<64, 64, 64, 1, 7, 1, -3, 0, [0-9]*, 1>;/using C_u0same = LdsLayer<64, 64, 64, 1, 7, 1, -3, 0, $2, 1>;/" \
      -e "s/using C_u1same = LdsLayer<32, 32, 32, 1, 7, 1, -3, 0, [0-9]*, 1>;/using C_u1same = LdsLayer<32, 32, 32, 1, 7, 1, -3, 0, $3, 1>;/" \
      -e "s/using C_d3same = LdsLayer<32, 0, 64, 1, 7, 1, -3, 0, [0-9]*, 1>;/using C_d3same = LdsLayer<32, 0, 64, 1, 7, 1, -3, 0, $4, 1>;/" \
      -e "s/using C_d2same = LdsLayer<16, 0, 32, 1, 7, 1, -3, 0, [0-9]*, 1>;/using C_d2same = LdsLayer<16, 0, 32, 1, 7, 1, -3, 0, $5, 1>;/" \
      -i $F
  make -C volpick_amd/csrc -j8 > /dev/null 2>&1
  echo "== $1 (u0same NB=$2 u1same NB=$3 d3same NB=$4 d2same NB=$5)"
  python tools/core_clock.py 2>&1 | grep -E "d2same|d3same|u0same|u1same|^total|whole window"
}
run base 3 6 3 3
run a 1 3 1 2
run b 2 2 2 1
run c 1 6 3 3
run base 3 6 3 3
