"""GPU box: the 1-GPU denominator of the strong-scaling leg -- ONE batch of BASELINE configs[2] (100 000 HiFi pairs) on one MI355X,
through bench.py itself -- written to profiles-style JSON (copy it to profiles/strong_1gpu.json; bench.py prints
strong.speedup_vs_recorded_1gpu against it at N > 1).      python tools/record_strong_1gpu.py OUT.json [pairs] [steps]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, pairs, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100000, int(sys.argv[3]) if len(sys.argv) > 3 else 10
p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "C2", "--scaling", "strong", "--pairs", str(pairs), "--steps", str(steps), "--warmup", "2",
                    "--no-cpu-baseline", "--no-gasal-api"], stdout=subprocess.PIPE, check=True)
line = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
rec = dict(config="C2", pairs=pairs, scoring="m2x4q4r2", ms_per_step=line["ms_per_step"], gcups=line["value"], kernel_ms=line["kernel_ms"], steps=steps,
           kernel=line["config"]["kernel"], schedule=line["config"]["preemptive_schedule_rank0"], int16_steps=line["config"]["int16_steps_rank0"],
           source="python bench.py --config C2 --scaling strong --pairs %d --steps %d --warmup 2 on one MI355X (tools/record_strong_1gpu.py)" % (pairs, steps))
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
