for sg in 0 4352 69888 1048832; do echo "== MULLS_STAGGER $sg"; MULLS_STAGGER=$sg timeout 200 python tools/gpu_modes.py 4096 | cut -c1-12,62-200; done
