# round 6, closing campaign on the final kernels (shipped library): new seeds, the z-walk route toggled per case in fuzz_hot
{ for sd in 9201 9202 9203 9204; do timeout 900 python tests/fuzz/fuzz_hot.py $sd 300; done
  timeout 900 python tests/fuzz/fuzz_parity.py 9211 400; timeout 600 python tests/fuzz/fuzz_int.py 9212 250;
  timeout 900 python tests/fuzz/fuzz_round4.py 9213 250; timeout 600 python tests/fuzz/fuzz_filter.py 9214 500; timeout 600 python tests/fuzz/fuzz_api.py 9215 300;
  FUZZ_FIELD_STRENGTH=strong timeout 900 python tests/fuzz/fuzz_hot.py 9216 300; } 2>&1 | grep -v amdgpu.ids | grep "cases\|FAIL\|err\|Error" > gpurun_out/r06_fuzz_closing.txt
cat gpurun_out/r06_fuzz_closing.txt
