set -u
bash profiles/run_profile.sh r02 > gpurun_out/r02_run_profile.log 2>&1
bash profiles/pmc_config.sh r02_c3 c3 back_pass_q4 > /dev/null 2>&1
bash profiles/pmc_config.sh r02_c4 c4 back_pass_mfma > /dev/null 2>&1
DDP_C4_LIMS=0.05 python profiles/bench_configs.py c4 > gpurun_out/r02_c4_lims.json 2>&1
tail -30 gpurun_out/r02_run_profile.log | cut -c1-300
cat gpurun_out/r02_c3/summary.txt | tail -32
cat gpurun_out/r02_c4/summary.txt | tail -34
cat gpurun_out/r02_c4_lims.json
