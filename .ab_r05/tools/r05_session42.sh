#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s42; mkdir -p $OUT
bash tools/profile_r05.sh r05final > $OUT/profile.log 2>&1
{
echo "--- the matrix in registers (FRANK_AMD_K2_RR=1)"; FRANK_AMD_K2_RR=1 python3 tools/k2_loaded.py 1 64 128 192 256
echo "--- the forms that work in memory (FRANK_AMD_K2_RR=0)"; FRANK_AMD_K2_RR=0 python3 tools/k2_loaded.py 1 64 128 192 256
} > $OUT/loaded_forms.txt 2>&1
{
echo "--- default (the form by load)"; python3 tools/steady_state2.py 1 4000
echo "--- FRANK_AMD_K2_RR=0"; FRANK_AMD_K2_RR=0 python3 tools/steady_state2.py 1 4000
echo "--- FRANK_AMD_K2_RR=1"; FRANK_AMD_K2_RR=1 python3 tools/steady_state2.py 1 4000
} 2>&1 | grep -v "^$" > $OUT/steady_forms.txt
FRANK_AMD_LIB=$PWD/frank_amd/libfrank_hip_timing.so FRANK_AMD_K2_RR=1 timeout 300 python3 tools/rr_trace.py > $OUT/rr_trace.txt 2>&1
{ tools/microbench/mfma_f64_bench; tools/microbench/tile_step_bench; } > $OUT/microbench.txt 2>&1
