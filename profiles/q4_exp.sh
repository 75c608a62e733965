export DDP_BC_STEPS=300 DDP_BC_WARMUP=50
# (the DDP_Q4_EXP removal experiments exist only in a profiling build:  bash profiles/build_variant.sh q4prof back_pass_q4.hip "-DDDP_PROFILE_BUILD";  export DDP_AMD_LIB=.../build/libddp_q4prof.so)
for e in 0 1 2 3 4 7 0; do
  DDP_Q4_EXP=$e python profiles/bench_configs.py c3 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('EXP=$e', d['back_pass_ms'], d['back_pass_ms_median'], d['back_pass_ms_min'])"
done
