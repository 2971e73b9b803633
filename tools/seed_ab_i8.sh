for n in 2048 4096 16384 65536; do echo "seed $n"; INNR_GEMM_SEED_N=$n python tools/run_c2.py i8 10 2>&1 | grep -v amdgpu; done
