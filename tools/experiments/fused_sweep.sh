echo "== consumers alone (producers idle, no waits), C=230"; timeout -k 10 120 python tools/fused_bwd_probe.py 786432 100 64 3 2>&1 | grep "^fused:"
echo "== consumers alone, C=128"; timeout -k 10 120 python tools/fused_bwd_probe.py 786432 500 64 3 2>&1 | grep "^fused:"
echo "== P=141 producers never wait, consumers do not wait"; timeout -k 10 120 python tools/fused_bwd_probe.py 786432 550 64 5 2>&1 | grep "^fused:"
echo "== P=230 producers never wait, consumers do not wait"; timeout -k 10 120 python tools/fused_bwd_probe.py 786432 900 64 5 2>&1 | grep "^fused:"
