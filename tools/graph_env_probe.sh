for e in "" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=64" "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "DEBUG_CLR_BLIT_KERNARG_OPT=1" "DEBUG_HIP_KERNARG_COPY_OPT=0"; do
  echo "== $e"; env $e python bench.py --mode train --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done
