#!/bin/bash
# Dev tool: renders the first frames of the reference's default animation (1080x720, 2500 spp, depth 50)
# with rtp_main; prints the reference's "frame \t ms \t rays" lines.
cd "$(dirname "$0")/.."
N=${1:-3}
./ray-tracing-practice_amd/rtp_main --default | sed "1s/.*/$N/; 2s#.*#/tmp/rtp_default_%d.png#" > /tmp/rtp_default_cfg.txt
echo "--- sequential driver"; ./ray-tracing-practice_amd/rtp_main --gpu < /tmp/rtp_default_cfg.txt
echo "--- pipelined driver"; ./ray-tracing-practice_amd/rtp_main --gpu --devices 1 < /tmp/rtp_default_cfg.txt
ls -la /tmp/rtp_default_0.png
