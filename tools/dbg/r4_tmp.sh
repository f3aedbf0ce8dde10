for P in -1 0 -1 0; do echo "pre priority $P"; IMS_FOCAL_PRE_PRIORITY=$P R4_SKIP_SINGLE=1 R4_CONC=4 python3 tools/dbg/r4_c5.py 189 2>&1 | grep concurrent; done
