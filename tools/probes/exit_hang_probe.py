"""Run a pytest selection in-process with (a) faulthandler's watchdog for the Python stacks and (b) a SIGUSR2 handler
(btsig.so) that makes every thread write its native backtrace: a hang in fork()/exit shows where each thread sits."""
import ctypes
import faulthandler
import os
import sys

import pytest

if __name__ == "__main__":   # multiprocessing's spawn re-imports the main module in its children
    here = os.path.dirname(os.path.abspath(__file__))
    out = open("gpurun_out/hang_trace.txt", "w")
    faulthandler.dump_traceback_later(int(sys.argv[1]), exit=False, file=out)
    try:
        bt = ctypes.CDLL(os.path.join(here, "btsig.so"))
        native = os.open("gpurun_out/hang_native.txt", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        bt.btsig_install(native)
    except OSError as e:
        print("no btsig.so:", e, flush=True)
    rc = pytest.main(sys.argv[2:])
    print("pytest returned", rc, flush=True)
    import threading

    print("threads alive:", [t.name for t in threading.enumerate()], flush=True)
    sys.exit(rc)
