for o in "" "warp_f32_px=1" "warp_f32_px=2"; do echo "== $o"; python scripts/diag/warp_channels_r06.py $o 2>&1 | grep -E "(warp_affine|warp_perspective|remap) f32.*(nearest|bilinear) c(1|3)"; done
