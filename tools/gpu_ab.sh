cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWITH_PROD -o /tmp/mv tools/experiments/matrix_variants.hip && /tmp/mv 32768 | grep -E "wide8|PROD"
