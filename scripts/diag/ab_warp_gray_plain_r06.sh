echo "== product"; python scripts/diag/warp_channels_r06.py 2>&1 | grep -E "u8 .*c1|u8 c1"
echo "== plain stores everywhere"; KORNIA_HIP_LIB=$PWD/scripts/ubench/bin/libkornia_hip_plain.so python scripts/diag/warp_channels_r06.py 2>&1 | grep -E "u8 .*c1|u8 c1"
