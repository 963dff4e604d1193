"""Driver of exit_hang_probe.py (touches no GPU itself): starts it as a child, writes a heartbeat, and if the child is
still alive after LIMIT seconds dumps the process tree (/proc state, wchan of every task), asks the child for native
backtraces (SIGUSR2) and then kills exactly the processes it found.
usage: hang_watch.py LIMIT pytest-args..."""
import os
import signal
import subprocess
import sys
import time

limit = int(sys.argv[1])
log = open("gpurun_out/hang_watch.log", "w")


def say(*a):
    print(*a, file=log, flush=True)


def stat(pid):
    try:
        s = open(f"/proc/{pid}/stat").read()
    except OSError:
        return None
    comm = s[s.index("(") + 1:s.rindex(")")]
    rest = s[s.rindex(")") + 2:].split()
    return comm, rest[0], int(rest[1])   # comm, state, ppid


def descendants(root):
    kids = {}
    for d in os.listdir("/proc"):
        if d.isdigit():
            st = stat(int(d))
            if st:
                kids.setdefault(st[2], []).append(int(d))
    out, todo = [], [root]
    while todo:
        p = todo.pop()
        out.append(p)
        todo += kids.get(p, [])
    return out


def dump(pid):
    st = stat(pid)
    say(f"--- pid {pid} {st}")
    try:
        say("cmdline:", open(f"/proc/{pid}/cmdline").read().replace("\0", " ")[:200])
    except OSError:
        pass
    try:
        tasks = sorted(os.listdir(f"/proc/{pid}/task"), key=int)
    except OSError:
        return
    for t in tasks:
        row = []
        for f in ("comm", "wchan", "syscall"):
            try:
                row.append(open(f"/proc/{pid}/task/{t}/{f}").read().strip())
            except OSError as e:
                row.append(f"<{f}: {e.strerror}>")
        try:
            s = open(f"/proc/{pid}/task/{t}/stat").read()
            row.append("state=" + s[s.rindex(")") + 2:].split()[0])
        except OSError:
            pass
        say(f"  tid {t}: " + " | ".join(row))


child = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "exit_hang_probe.py"),
                          str(limit - 20)] + sys.argv[2:])
t0 = time.time()
while child.poll() is None and time.time() - t0 < limit:
    time.sleep(10)
    say(f"heartbeat {time.time() - t0:.0f}s")
if child.poll() is not None:
    say("child exited with", child.returncode)
    sys.exit(child.returncode)
say("STILL ALIVE after", limit, "s")
tree = descendants(child.pid)
for p in tree:
    dump(p)
os.kill(child.pid, signal.SIGUSR2)
time.sleep(5)
for p in tree:
    dump(p)
for p in reversed(tree):
    try:
        os.kill(p, signal.SIGKILL)
    except OSError:
        pass
say("killed", tree)
sys.exit(3)
