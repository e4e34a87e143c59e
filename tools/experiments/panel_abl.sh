(timeout 600 python -m pytest tests/test_panel_gpu.py -q 2>&1 | tail -5) > gpurun_out/panel_test.log
for v in "" _abl1 _abl2 _abl4 _abl6 _abl8 _abl14; do echo "== variant $v"; SMX_LIB=summarymixing_amd/libsmx$v.so D=512 F=2048 timeout 200 python tools/panel_bench.py 2>&1 | grep " + dropout\|bias only"; SMX_LIB=summarymixing_amd/libsmx$v.so D=256 F=1024 timeout 200 python tools/panel_bench.py 2>&1 | grep " + dropout"; done > gpurun_out/panel_abl.log 2>&1
cat gpurun_out/panel_test.log gpurun_out/panel_abl.log
