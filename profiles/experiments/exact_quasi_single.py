"""Single frames of exactly periodic plans with a source step > 1 (quasi-periodic kernel, EXACT variant) against the gather kernel."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
EXTRA = {"X54": ("Y8", 1536, 864, 1920, 1080, dict(tap=3), 1), "X43T4": ("Y8", 1440, 1080, 1920, 1440, dict(tap=4), 1),
         "X85": ("Y8", 1200, 675, 1920, 1080, dict(tap=3), 1), "X32S": ("Y8", 640, 360, 960, 540, dict(tap=3), 1)}
bench.CONFIGS.update(EXTRA)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.argv = [sys.argv[0]] + sys.argv[2:]
    bench.main()
    sys.exit(0)
for c in ["U43"] + list(EXTRA):
    for f in (1, 2):
        for m in (0, 1):
            r = subprocess.run([sys.executable, __file__, "--child", "--config", c, "--frames", str(f), "--steps", "200", "--warmup", "20",
                                "--no-cpu-baseline", "--kernel-mode", str(m)], capture_output=True, text=True)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not lines:
                print(c, f, m, "failed", r.stderr[-200:]); continue
            d = json.loads(lines[-1])
            print("%-6s frames=%d mode=%d %-22s %7.1f Gpix/s step %.4f ms" % (c, f, m, d["config"]["kernel"], d["value"] / 1e3, d["ms_per_step"]))
