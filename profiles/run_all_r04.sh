set -u
# everything behind the round-4 numbers in ONE gpurun call: bench line + kernel statistics + PMC traffic of the headline (run_profile.sh:
# the headline's backward kernel is now sh_back_kernel), un-profiled event times, warm kernel statistics, PMC traffic and SQ counters of
# the other configs (C3 / C5 on the re-laid-out chunk kernel back_pass_q4c, C5 without its prepass), the shared-LTI A/B over batch sizes,
# whole solves incl. the slot scheduler.
mkdir -p gpurun_out
bash profiles/run_profile.sh r04 > gpurun_out/r04_run_profile.log 2>&1
bash profiles/pmc_config.sh r04_c3 c3 back_pass_q4c,forward_pend_row_kernel,df_pendcart_kernel > /dev/null 2>&1
bash profiles/pmc_config.sh r04_c2tv c2tv back_pass_mx,forward_pipe > /dev/null 2>&1
bash profiles/pmc_config.sh r04_c4 c4 back_pass_mfma,forward_big64 > /dev/null 2>&1
bash profiles/pmc_config.sh r04_c5 c5 back_pass_q4c,forward_pend_row_kernel,fcov_q4l_kernel,kl_div_lds_kernel > /dev/null 2>&1
DDP_C4_LIMS=0.05 DDP_C4_SOLVE=0 DDP_BC_STEPS=40 DDP_BC_WARMUP=8 python profiles/bench_configs.py c4 > gpurun_out/r04_c4_lims.json 2>&1
python profiles/ab_sh.py 1024 2048 4096 8192 32768 > gpurun_out/r04_shared_lti_ab.txt 2>&1
(python profiles/ilqg_c2.py; python profiles/ilqg_c3.py; python profiles/ilqgkl_c5.py; python profiles/host_io_rate.py; python profiles/ilqg_queue_c3.py) 2>&1 | grep -E "^C[235]|GPU phases|iterations per|live traj|host-pointer pass|^queue|^lock step|same summaries" > gpurun_out/r04_solves.txt
tail -30 gpurun_out/r04_run_profile.log | cut -c1-300
for c in c3 c2tv c4 c5; do cat gpurun_out/r04_$c/summary.txt | tail -60; done
cat gpurun_out/r04_c4_lims.json gpurun_out/r04_shared_lti_ab.txt gpurun_out/r04_solves.txt
