"""Developer tool: copy the counter files of tools/collect_r04.sh (gpurun_out/r04c/) into profiles/ under the names bench.py reads."""
import json, os
src = 'gpurun_out/r04c'
cmds = {'bench_traffic_pmc.json': ('r04_traffic_pmc.json', 'bench.py --steps 50 --warmup 10 --no-cpu-baseline'),
        'angles180_traffic_pmc.json': ('r04_angles180_traffic_pmc.json', 'bench.py --steps 50 --warmup 10 --no-cpu-baseline --angles 180'),
        'angles180_compact_traffic_pmc.json': ('r04_angles180_compact_traffic_pmc.json', 'bench.py --steps 50 --warmup 10 --no-cpu-baseline --angles 180 --plan-format compact'),
        'n512_traffic_pmc.json': ('r04_n512_traffic_pmc.json', 'bench.py --no-cpu-baseline --mode n512'),
        'training_call_traffic_pmc.json': ('r04_training_call_traffic_pmc.json', 'tools/trace_training_call.py 50')}
for a, (b, cmd) in cmds.items():
    if os.path.exists(os.path.join(src, a)):
        d = json.load(open(os.path.join(src, a)))
        d['_command'] = cmd + ' (tools/collect_r04.sh traffic)'
        json.dump(d, open(os.path.join('profiles', b), 'w'), indent=1)
for a, b, cmd in (('sq_a20_counters.json', 'r04_sq_a20_counters.json', ''), ('sq_angles180_counters.json', 'r04_sq_angles180_counters.json', '--angles 180'),
                  ('sq_n512_counters.json', 'r04_sq_n512_counters.json', '--mode n512')):
    if os.path.exists(os.path.join(src, a)):
        d = json.load(open(os.path.join(src, a)))
        d['_how'] = ('tools/collect_sq.sh: three rocprofv3 --pmc passes over bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph ' + cmd +
                     '; means per dispatch, summed over the chip (256 CUs, 32 shader engines)')
        json.dump(d, open(os.path.join('profiles', b), 'w'), indent=1, sort_keys=True)
