L=("m.s3.b.conv2 192->192 3x3 @40" "m.s4.b.conv2 384->384 3x3 @20" "m.s2.b.conv2 96->96 3x3 @80" "s3.b.conv2 128->128 3x3 @40" "s2.b.conv2 64->64 3x3 @80" "s4.b.conv2 256->256 3x3 @20" "s1.b.conv2 32->32 3x3 @160")
for v in "KODHIP_WGRAD_ROW3=0" "KODHIP_WGRAD_ROW3=2" "KODHIP_WGRAD_ROW3=2 KODHIP_WGRAD_ROW3_SLOTS=1536" "KODHIP_WGRAD_ROW3=2 KODHIP_WGRAD_ROW3_SLOTS=768" "KODHIP_WGRAD_ROW3=0 KODHIP_WGRAD_SLOTS=256"; do
  echo "== [$v]"
  env $v BENCH_CONV_ONLY=wgrad timeout -k 10 200 python tools/bench_conv.py "${L[@]}" 2>&1 | grep -v amdgpu.ids | cut -c1-90
done
