cd $GRAFT_REPO_ROOT
(python tools/stage_profile.py --batch 5120 --workload sphere --steps 60 > /tmp/sp.log 2>&1; python tools/stage_profile.py --batch 5120 --workload static --steps 30 >> /tmp/sp.log 2>&1) &
PID=$!
for i in $(seq 1 40); do
  sleep 1.5
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|memory)" | tr -s ' ' | tr '\n' '|' | cut -c1-400; echo
  kill -0 $PID 2>/dev/null || break
done
wait $PID
grep workload /tmp/sp.log
