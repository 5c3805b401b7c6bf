set -e
cd $GRAFT_REPO_ROOT
run() { # name d1same u1same u2same
  sed -e "s/using C_d1same = LdsLayer<8, 0, 16, 1, 7, 1, -3, 0, [0-9]*, 1>;/using C_d1same = LdsLayer<8, 0, 16, 1, 7, 1, -3, 0, $2, 1>;/" \
      -e "s/using C_u1same = LdsLayer<32, 32, 32, 1, 7, 1, -3, 0, [0-9]*, 1>;/using C_u1same = LdsLayer<32, 32, 32, 1, 7, 1, -3, 0, $3, 1>;/" \
      -e "s/using C_u2same = LdsLayer<16, 16, 16, 1, 7, 1, -3, 0, [0-9]*, 1>;/using C_u2same = LdsLayer<16, 16, 16, 1, 7, 1, -3, 0, $4, 1>;/" \
      -i volpick_amd/csrc/phasenet_fused.hip
  make -C volpick_amd/csrc -j8 > /dev/null 2>&1
  echo "== $1 (d1same NB=$2 u1same NB=$3 u2same NB=$4)"
  python tools/core_clock.py 2>&1 | grep -E "d1same|u1same|u2same|total"
}
run base 6 6 6
run nb4 4 4 4
run nb8 8 6 8
run nb3 3 3 3
