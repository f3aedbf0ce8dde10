for J in 8 0; do
echo "== joint $J, 96 CCDs"
IMS_FOCAL_JOINT=$J python3 tools/dbg/r4_joint_host.py 96 2>&1 | cut -c1-150 | grep -v "^$" | head -32
done
