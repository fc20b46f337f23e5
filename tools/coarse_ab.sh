for v in "$@"; do echo "== $v"; DDK_LIB=$(pwd)/ab_libs/libddk_$v.so timeout 200 python tools/conv_trace.py --coarse --layer 3 2>&1 | grep "wave [04]:"; done
