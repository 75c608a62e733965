set -u
# everything behind the round-3 numbers in ONE gpurun call: bench line + kernel statistics + PMC traffic of the headline (run_profile.sh),
# then un-profiled event times, warm kernel statistics, PMC traffic and SQ counters of every other config and of the kernels that had none
mkdir -p gpurun_out
bash profiles/run_profile.sh r03 > gpurun_out/r03_run_profile.log 2>&1
bash profiles/pmc_config.sh r03_c3 c3 back_pass_q4,forward_pend_row_kernel,df_pendcart_kernel > /dev/null 2>&1
bash profiles/pmc_config.sh r03_c2tv c2tv back_pass_mx,forward_pipe_kernel > /dev/null 2>&1
bash profiles/pmc_config.sh r03_c4 c4 back_pass_mfma,forward_big64 > /dev/null 2>&1
bash profiles/pmc_config.sh r03_c5 c5 back_pass_q4l_kernel,gps_combine_kernel,forward_pend_row_kernel,fcov_q4l_kernel,kl_div_lds_kernel > /dev/null 2>&1
DDP_C4_LIMS=0.05 DDP_C4_SOLVE=0 DDP_BC_STEPS=40 DDP_BC_WARMUP=8 python profiles/bench_configs.py c4 > gpurun_out/r03_c4_lims.json 2>&1
(python profiles/ilqg_c2.py; python profiles/ilqg_c3.py; python profiles/ilqgkl_c5.py; python profiles/host_io_rate.py) 2>&1 | grep -E "^C[235]|GPU phases|iterations per|live traj|host-pointer pass" > gpurun_out/r03_solves.txt
tail -30 gpurun_out/r03_run_profile.log | cut -c1-300
for c in c3 c2tv c4 c5; do cat gpurun_out/r03_$c/summary.txt | tail -60; done
cat gpurun_out/r03_c4_lims.json
cat gpurun_out/r03_solves.txt
# machine-filling batch: removal experiments of back_pass_dppw and the batch cross-over of the three LTI backward kernels
bash profiles/ab_dppw_exp.sh > gpurun_out/r03_fill_dppw_exp.txt 2>&1
bash profiles/ab_fill_crossover.sh > gpurun_out/r03_fill_crossover.txt 2>&1
cat gpurun_out/r03_fill_dppw_exp.txt gpurun_out/r03_fill_crossover.txt
