cd "$GRAFT_REPO_ROOT"
for c in "X=0" "ARUCOHIP_QUAD_BLOCKS=16" "ARUCOHIP_QUAD_BLOCKS=32" "ARUCOHIP_QUAD_BLOCKS=48" "ARUCOHIP_QUAD_BLOCKS=12" "ARUCOHIP_CAND_WAVES=64" "ARUCOHIP_FORK_AFTER=2" "ARUCOHIP_FORK_AFTER=4"; do
 env $c python tools/kbench.py --frames 512 --steps 6 --tag "$c" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms']; print(j['tag'], 'cand', k['candidates_kernel'], 'walker', k['walker_kernel'], 'long', k['walker_long_kernel'], 'quad', k['contour_quad_kernel'], 'total', j['total_ms'], j['markers'])"
done
