"""Which fork / join patterns HIP's stream capture (ROCm 7.2, torch.cuda.graph) survives: each pattern runs eagerly, is captured, replayed, in a process of its own.
Measured on MI355X: a forked (non-origin) stream that already holds captured work and then waits for an event recorded on a forked stream -- itself (`self_wait`),
or a stream it forked (`nested`, `flat`) -- makes hipStreamEndCapture fault (SIGSEGV); forked streams that only ever wait BEFORE their first captured launch
(`c_waits_b`, `b_joins_c_only`, `two_branches`) and self-waits on the origin stream are fine.  Consequence (device/agent57_fast.py): a network's update, which forks
its weight-gradient stream, cannot run on a forked lane of the update's graph."""
import sys, subprocess, textwrap
PAT = {
"nested": """
e0.record(cur); b.wait_event(e0)
with torch.cuda.stream(b):
    y = x * 2
    e1.record(b); c.wait_event(e1)
    with torch.cuda.stream(c):
        z = y + 1
        e2.record(c)
    w = y * 3
    b.wait_event(e2)
    w = w + z
    e3.record(b)
v = x + 5
cur.wait_event(e3)
out = v + w
""",
"flat": """
e0.record(cur); b.wait_event(e0); c.wait_event(e0)
with torch.cuda.stream(b):
    y = x * 2
    e1.record(b); c.wait_event(e1)
    with torch.cuda.stream(c):
        z = y + 1
        e2.record(c)
    w = y * 3
    b.wait_event(e2)
    w = w + z
    e3.record(b)
v = x + 5
cur.wait_event(e3)
out = v + w
""",
"self_wait": """
e0.record(cur); b.wait_event(e0)
with torch.cuda.stream(b):
    y = x * 2
    e1.record(b); b.wait_event(e1)
    w = y * 3
    e3.record(b)
v = x + 5
cur.wait_event(e3)
out = v + w
""",
"c_waits_b": """
e0.record(cur); b.wait_event(e0); c.wait_event(e0)
with torch.cuda.stream(b):
    w = x * 2
    e1.record(b)
with torch.cuda.stream(c):
    c.wait_event(e1)
    z = w * 3
    e2.record(c)
v = x + 5
cur.wait_event(e1); cur.wait_event(e2)
out = v + w + z
""",
"b_joins_c_only": """
e0.record(cur); b.wait_event(e0); c.wait_event(e0)
with torch.cuda.stream(b):
    w = x * 2
    e1.record(b)
with torch.cuda.stream(c):
    c.wait_event(e1)
    z = w * 3
    e2.record(c)
v = x + 5
cur.wait_event(e2)
out = v + z
""",
"main_self_wait": """
y = x * 2
e1.record(cur); cur.wait_event(e1)
out = y * 3
""",
"two_branches": """
e0.record(cur); b.wait_event(e0); c.wait_event(e0)
with torch.cuda.stream(b):
    w = x * 2
    e1.record(b)
with torch.cuda.stream(c):
    z = x * 3
    e2.record(c)
v = x + 5
cur.wait_event(e1); cur.wait_event(e2)
out = v + w + z
""",
}
if len(sys.argv) > 1:
    import torch
    x = torch.ones(1024, device="cuda")
    b, c = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
    e0, e1, e2, e3 = (torch.cuda.Event() for _ in range(4))
    body = PAT[sys.argv[1]]
    cur = torch.cuda.current_stream()
    exec(body)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            exec(body)
    g.replay(); torch.cuda.synchronize()
    print(sys.argv[1], "ok", float(out[0]))
else:
    for k in PAT:
        r = subprocess.run([sys.executable, __file__, k], capture_output=True, text=True)
        print(k, r.returncode, r.stdout.strip()[-60:], r.stderr.strip()[-200:] if r.returncode else "")
