set -u
# everything behind the round-5 numbers in ONE gpurun call: the GPU test suite, bench line + kernel statistics + PMC traffic of the headline
# (run_profile.sh), the floor microbenchmark of the shared chain, kernel statistics / PMC of the off-shape configs (wide tile kernel, padded
# row kernels, mid-size matrix-core kernels) and of C3 / C4 / C5, the mid-size sweep, whole solves.  (The A/B of the shared chain against its
# round-4 form and its phase profile need build/libddp_shold.so / libddp_shprof.so: profiles/build_variant.sh, profiles/r05_sh_ab.sh.)
mkdir -p gpurun_out
(python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/r05_tests_full.txt
bash profiles/run_profile.sh r05 > gpurun_out/r05_run_profile.log 2>&1
[ -x ./profiles/microbench/sh_chain_floor ] && ./profiles/microbench/sh_chain_floor > gpurun_out/r05_sh_chain_floor.txt 2>&1
bash profiles/pmc_config.sh r05_offA offA back_pass_mxg,forward_row > /dev/null 2>&1
bash profiles/pmc_config.sh r05_offB offB back_pass_row,forward_row > /dev/null 2>&1
bash profiles/pmc_config.sh r05_offC offC back_pass_mid,forward_mid > /dev/null 2>&1
bash profiles/pmc_config.sh r05_offD offD back_pass_mid,forward_mid > /dev/null 2>&1
bash profiles/pmc_config.sh r05_offL offL back_pass_mxg,forward_dpp > /dev/null 2>&1
bash profiles/pmc_config.sh r05_c3 c3 back_pass_q4c,forward_pend_row_kernel,df_pendcart_kernel > /dev/null 2>&1
bash profiles/pmc_config.sh r05_c4 c4 back_pass_mfma,forward_big64 > /dev/null 2>&1
bash profiles/pmc_config.sh r05_c5 c5 back_pass_q4c,forward_pend_row_kernel,fcov_q4l_kernel,kl_div_lds_kernel > /dev/null 2>&1
bash profiles/pmc_config.sh r05_c2tv c2tv back_pass_mx,forward_pipe > /dev/null 2>&1
bash profiles/r05_mid_sweep.sh > gpurun_out/r05_mid_sweep.txt 2>&1
(python profiles/ilqg_c2.py; python profiles/ilqg_c3.py; python profiles/ilqgkl_c5.py; python profiles/ilqg_queue_c3.py) 2>&1 | grep -E "^C[235]|GPU phases|iterations per|live traj|^queue|^lock step|same summaries" > gpurun_out/r05_solves.txt
cat gpurun_out/r05_tests_full.txt; tail -25 gpurun_out/r05_run_profile.log | cut -c1-400
for c in offA offB offC offD offL c3 c4 c5 c2tv; do head -4 gpurun_out/r05_$c/summary.txt | cut -c1-700; done
cat gpurun_out/r05_mid_sweep.txt gpurun_out/r05_solves.txt
