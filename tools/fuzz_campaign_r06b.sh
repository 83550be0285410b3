# round 6, second closing campaign (final kernels, new seeds; fuzz_hot toggles the z-walk route per case)
{ for sd in 9301 9302 9303 9304 9305 9306; do timeout 900 python tests/fuzz/fuzz_hot.py $sd 350; done
  for sd in 9311 9312; do timeout 900 python tests/fuzz/fuzz_parity.py $sd 400; done
  timeout 900 python tests/fuzz/fuzz_int.py 9321 400; timeout 900 python tests/fuzz/fuzz_round4.py 9322 300;
  timeout 600 python tests/fuzz/fuzz_api.py 9323 400; timeout 600 python tests/fuzz/fuzz_filter.py 9324 500;
  FUZZ_FIELD_STRENGTH=strong timeout 900 python tests/fuzz/fuzz_hot.py 9325 400; } 2>&1 | grep -v amdgpu.ids | grep "cases\|FAIL\|err\|Error" > gpurun_out/r06_fuzz_closing2.txt
cat gpurun_out/r06_fuzz_closing2.txt
