for m in ${MODES:-0 1 2 3}; do LMAT_K4_MODE=$m python bench.py --no-cpu --no-e2e 2>&1 >/dev/null | grep "timed region" | sed "s/^/mode=$m /"; done
