mkdir -p gpurun_out/r6h
python tools/flavour_table.py 10 > gpurun_out/r6h/flavour_halfcheetah.txt 2>&1
python tools/flavour_table.py 6 - slim_humanoid > gpurun_out/r6h/flavour_slim_humanoid.txt 2>&1
grep -v amdgpu gpurun_out/r6h/flavour_halfcheetah.txt; grep -v amdgpu gpurun_out/r6h/flavour_slim_humanoid.txt
