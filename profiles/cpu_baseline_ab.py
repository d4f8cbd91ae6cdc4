"""A/B of the CPU layer's allocator setting (run on the GPU box's host): python profiles/cpu_baseline_ab.py"""
import json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
code = "import sys, json; sys.path.insert(0, %r); import cpu_layer as c; print(json.dumps(c.measure(chi=32, L=8)))" % os.path.join(here, "..", "oracle")
for tag, env in (("mallopt", {}), ("default", {"TNQS_CPU_NO_MALLOPT": "1"})):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print(tag, out.stdout.strip() or out.stderr[-400:])
