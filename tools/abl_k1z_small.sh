# dev tool: the z-walk route on small volumes with the shipped strip rule: strong hint against the default routing, by sigma
for sg in 5 10 15; do for sh in 128x128x128 192x192x192 96x104x120 16x128x128x128; do
  echo -n "sigma $sg auto   "; SIGMA=$sg python tools/time_fwd_shape.py $sh 2>&1 | grep -v amdgpu | cut -c1-46
  echo -n "sigma $sg strong "; EDHIP_FIELD_STRENGTH=strong SIGMA=$sg python tools/time_fwd_shape.py $sh 2>&1 | grep -v amdgpu | cut -c1-46
done; done
python -m pytest tests/test_zwalk_route.py -x -q -m gpu 2>&1 | tail -2
