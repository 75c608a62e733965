set -u
bash profiles/run_profile.sh r02 > gpurun_out/r02_run_profile.log 2>&1
bash profiles/pmc_config.sh r02_c3 c3 back_pass_q4 > /dev/null 2>&1
bash profiles/pmc_config.sh r02_c4 c4 back_pass_mfma > /dev/null 2>&1
DDP_C4_LIMS=0.05 DDP_C4_SOLVE=0 DDP_BC_STEPS=40 DDP_BC_WARMUP=8 python profiles/bench_configs.py c4 > gpurun_out/r02_c4_lims.json 2>&1
(python profiles/ilqg_c2.py; python profiles/ilqg_c3.py; python profiles/ilqgkl_c5.py) 2>&1 | grep -E "^C[235]|GPU phases|iterations per|live traj" > gpurun_out/r02_solves.txt
tail -30 gpurun_out/r02_run_profile.log | cut -c1-300
cat gpurun_out/r02_c3/summary.txt | tail -32
cat gpurun_out/r02_c4/summary.txt | tail -34
cat gpurun_out/r02_c4_lims.json
cat gpurun_out/r02_solves.txt
