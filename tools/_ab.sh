for r in 1 2 3; do for L in tt256 tt512; do echo "lib $L"; VOLPICK_HIP_LIB=$PWD/tools/_exp/lib_$L.so python tools/train_probe.py 2>&1 | tail -1; done; done
