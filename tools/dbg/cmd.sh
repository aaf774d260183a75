cd /root/repo
python tools/dbg/graph_host_time.py 2>&1 | grep capture_steps
