set -u
# everything behind the round-6 numbers, in parts so that no single gpurun call is long:  bash profiles/run_all_r06.sh <part>
#   a  GPU test suite, bench line + kernel statistics + PMC traffic of the headline (run_profile.sh), chain-floor microbenchmarks
#   b  kernel statistics / PMC of C4 (a2 layout), C4TV (a3 layout), offE (n = 48, m = 6): the run-time-sized matrix-core kernel back_pass_mf2
#   c  C3, C5, C2TV, offL, offB
#   d  offA, offC, offD, mid-size sweep, whole solves, phase profile of back_pass_mf2 (needs build/libddp_mf2prof.so: profiles/build_variant.sh)
mkdir -p gpurun_out
PART=${1:-a}
case $PART in
a)
  (python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/r06_tests_full.txt
  bash profiles/run_profile.sh r06 > gpurun_out/r06_run_profile.log 2>&1
  { ./profiles/microbench/sh_chain_floor; echo; ./profiles/microbench/q4c_chain_floor; echo; ./profiles/microbench/pend_row_chain_floor; } > gpurun_out/r06_floors.txt 2>&1
  cat gpurun_out/r06_tests_full.txt; tail -25 gpurun_out/r06_run_profile.log | cut -c1-400 ;;
b)
  bash profiles/pmc_config.sh r06_c4 c4 back_pass_mf2,forward_big64 > /dev/null 2>&1
  bash profiles/pmc_config.sh r06_c4tv c4tv back_pass_mf2,forward_big64 > /dev/null 2>&1
  bash profiles/pmc_config.sh r06_offE offE back_pass_mf2,forward_big > /dev/null 2>&1
  bash profiles/pmc_config.sh r06_c4old c4 back_pass_mfma,forward_big64 DDP_BACKPASS=old > /dev/null 2>&1
  for c in c4 c4tv offE c4old; do head -4 gpurun_out/r06_$c/summary.txt | cut -c1-900; grep "^## back_pass" gpurun_out/r06_$c/summary.txt; grep "SQ_LDS_BANK_CONFLICT\|SQ_ACTIVE_INST_LDS" gpurun_out/r06_$c/summary.txt | head -2; done ;;
c)
  bash profiles/pmc_config.sh r06_c3 c3 back_pass_q4c,forward_pend_row_kernel,df_pendcart_kernel > /dev/null 2>&1
  bash profiles/pmc_config.sh r06_c5 c5 back_pass_q4c,forward_pend_row_kernel,fcov_q4l_kernel,kl_div_lds_kernel > /dev/null 2>&1
  bash profiles/pmc_config.sh r06_c2tv c2tv back_pass_mx,forward_pipe > /dev/null 2>&1
  bash profiles/pmc_config.sh r06_offL offL back_pass_mxg,forward_dpp > /dev/null 2>&1
  bash profiles/pmc_config.sh r06_offB offB back_pass_row,forward_row > /dev/null 2>&1
  for c in c3 c5 c2tv offL offB; do head -4 gpurun_out/r06_$c/summary.txt | cut -c1-700; grep "^## back_pass" gpurun_out/r06_$c/summary.txt; done ;;
d)
  bash profiles/pmc_config.sh r06_offA offA back_pass_mxg,forward_row > /dev/null 2>&1
  bash profiles/pmc_config.sh r06_offC offC back_pass_mid,forward_mid > /dev/null 2>&1
  bash profiles/pmc_config.sh r06_offD offD back_pass_mid,forward_mid > /dev/null 2>&1
  bash profiles/r05_mid_sweep.sh > gpurun_out/r06_mid_sweep.txt 2>&1
  (python profiles/ilqg_c2.py; python profiles/ilqg_c3.py; python profiles/ilqgkl_c5.py; python profiles/ilqg_queue_c3.py) 2>&1 | grep -E "^C[235]|GPU phases|iterations per|live traj|^queue|^lock step|same summaries" > gpurun_out/r06_solves.txt
  R=$PWD/differentialdynamicprogramming.jl_amd/build
  if [ -f $R/libddp_mf2prof.so ]; then
    { echo "== C4 (n=64 m=8 N=256 B=1024), s_memtime ticks per step and wave"; DDP_C4_SOLVE=0 DDP_BC_WARMUP=1 DDP_BC_STEPS=4 DDP_AMD_LIB=$R/libddp_mf2prof.so timeout 300 python profiles/bench_configs.py c4 2>&1 | grep "PROF" | tail -4 | sort
      echo "== offE (n=48 m=6 N=300 B=1024)"; DDP_BC_WARMUP=1 DDP_BC_STEPS=4 DDP_AMD_LIB=$R/libddp_mf2prof.so timeout 300 python profiles/bench_configs.py offE 2>&1 | grep "PROF" | tail -4 | sort; } > gpurun_out/r06_mf2_phases.txt 2>&1
  fi
  for c in offA offC offD; do head -4 gpurun_out/r06_$c/summary.txt | cut -c1-700; grep "^## back_pass" gpurun_out/r06_$c/summary.txt; done
  cat gpurun_out/r06_mid_sweep.txt gpurun_out/r06_solves.txt gpurun_out/r06_mf2_phases.txt ;;
esac
