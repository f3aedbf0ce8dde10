drv() { env R4_SKIP_SINGLE=1 R4_CONC=4 "$@" python3 tools/dbg/r4_c5.py 189 2>&1 | grep concurrent; }
echo "init pre + fft mid"; drv IMS_FOCAL_JOINT_INIT=pre IMS_FOCAL_FFT=mid
echo "init pre + fft bulk"; drv IMS_FOCAL_JOINT_INIT=pre IMS_FOCAL_FFT=bulk
echo "init pre"; drv IMS_FOCAL_JOINT_INIT=pre
echo "init pre + fft mid again"; drv IMS_FOCAL_JOINT_INIT=pre IMS_FOCAL_FFT=mid
echo "default"; drv IMS_X=1
