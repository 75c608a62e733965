# result stores of the shared-LTI backward pass, non-temporal below DDP_SH_NT_MAX_B trajectories and plain from there on: where is the cross-over?
for r in 1 2; do
for t in 1000000 1 default; do
  if [ $t = default ]; then unset DDP_SH_NT_MAX_B; else export DDP_SH_NT_MAX_B=$t; fi
  echo "== DDP_SH_NT_MAX_B=$t ($( [ $t = 1 ] && echo plain || ([ $t = default ] && echo "default: nt below 3072" || echo nt) ))"; python profiles/ab_sh.py 1024 2048 3072 4096 8192 16384 32768 2>&1 | grep "shared  "
done; done
