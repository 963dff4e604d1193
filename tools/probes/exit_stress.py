"""Start exit_child.py K times, one after the other; a child still alive LIMIT seconds after it said 'child done' is
dumped (/proc state of every task, native backtraces via SIGUSR2) and killed.  usage: exit_stress.py K [LIMIT]"""
import os
import signal
import subprocess
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
K = int(sys.argv[1])
LIMIT = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
log = open("gpurun_out/exit_stress.log", "w")


def say(*a):
    print(*a, file=log, flush=True)


def dump(pid):
    try:
        tasks = sorted(os.listdir(f"/proc/{pid}/task"), key=int)
    except OSError:
        return
    seen = {}
    for t in tasks:
        row = []
        for f in ("comm", "wchan"):
            try:
                row.append(open(f"/proc/{pid}/task/{t}/{f}").read().strip())
            except OSError as e:
                row.append(f"<{e.strerror}>")
        key = " | ".join(row)
        seen.setdefault(key, []).append(t)
    for key, ts in seen.items():
        say(f"  {len(ts)} x {key}  (tids {ts[:4]}...)")


hung = 0
for i in range(K):
    native = f"gpurun_out/exit_native_{i}.txt"
    t0 = time.time()
    child = subprocess.Popen([sys.executable, os.path.join(here, "exit_child.py"), native, str(i)], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True)
    line = child.stdout.readline()
    while line and "child done" not in line:
        line = child.stdout.readline()
    t1 = time.time()
    try:
        child.wait(timeout=LIMIT)
        say(f"run {i}: work {t1 - t0:.1f}s exit {time.time() - t1:.2f}s rc={child.returncode}")
        os.unlink(native)
    except subprocess.TimeoutExpired:
        hung += 1
        say(f"run {i}: STILL ALIVE {LIMIT}s after 'child done'")
        dump(child.pid)
        os.kill(child.pid, signal.SIGUSR2)
        time.sleep(3)
        os.kill(child.pid, signal.SIGKILL)
        child.wait()
say("hung", hung, "of", K)
sys.exit(1 if hung else 0)
