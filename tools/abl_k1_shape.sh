# dev tool: K1z against the round-5 K1 (tools/libedhip_nok1z.so) on non-cubic / non-power-of-two shapes and batches, sigma 5 / 10 / 15 (scaled by extent / 256)
cp elasticdeform_amd/libedhip.so /tmp/ship.so
S="256x256x256 128x128x128 192x192x192 320x320x320 256x264x256 16x128x128x128 4x256x256x256"
for sg in 10 15; do for lib in /tmp/ship.so tools/libedhip_nok1z.so; do cp $lib elasticdeform_amd/libedhip.so; echo "== sigma $sg $lib"; SIGMA=$sg python tools/time_fwd_shape.py $S 2>&1 | grep -v amdgpu | cut -c1-46; done; done
cp /tmp/ship.so elasticdeform_amd/libedhip.so
