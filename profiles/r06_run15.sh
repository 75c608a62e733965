cd "$GRAFT_REPO_ROOT/profiles/microbench"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../differentialdynamicprogramming.jl_amd/csrc pend_row_chain_floor.hip -o /tmp/pend_floor 2>/dev/null
/tmp/pend_floor
