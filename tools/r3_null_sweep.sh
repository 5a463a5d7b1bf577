cd $GRAFT_REPO_ROOT
ln -sf /dev/null /dev/shm/null_r3.fq
python tests/manual/e2e_threads.py 1500000 2 'OUT=/dev/shm/null_r3.fq' 'OUT=/dev/shm/null_r3.fq TGSF_CTX_PER_DEVICE=4' 'OUT=/dev/shm/null_r3.fq TGSF_CTX_PER_DEVICE=5' 'OUT=/dev/shm/null_r3.fq TGSF_CTX_PER_DEVICE=6' 'OUT=/dev/shm/null_r3.fq TGSF_CTX_PER_DEVICE=4 TGSF_SCAN_THREADS=12' 'TGSF_CTX_PER_DEVICE=4' > gpurun_out/r3_null_ctx.txt 2>&1; cat gpurun_out/r3_null_ctx.txt
rm -f /dev/shm/null_r3.fq
