"""Where does the wall time of peng_motif go outside main()?  The bench's FASTA from a python parent: time until the
child's first stderr line (main has started and the clock line "[timing] start" is printed when PENGK_TIMING_START is set),
until its "[timing] total" line, until the process has gone.   usage (GPU box): python tools/e2e_parent_probe.py"""
import os, subprocess, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "peng-motif_amd", "host", "peng_motif")
tmp = tempfile.mkdtemp(prefix="pengk_probe_", dir="/dev/shm")
fa = os.path.join(tmp, "s.fa")
subprocess.run([os.path.join(ROOT, "tools", "synth_fasta"), fa, "10000000", "200", "1", "0"], check=True)
for how in ("pipe", "file", "pipe", "file"):
    for _ in range(2):
        t0 = time.perf_counter()
        if how == "pipe":
            p = subprocess.Popen([exe, fa, "-w", "10", "-o", os.path.join(tmp, "o.meme")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                 env=dict(os.environ, PENGK_TIMING="1"))
            first = total = None
            for line in p.stderr:
                now = time.perf_counter() - t0
                if first is None:
                    first = now
                if line.startswith(b"[timing] total"):
                    total = now
            p.wait()
            print("stderr to a pipe: first line %.3f  total line %.3f  gone %.3f" % (first, total, time.perf_counter() - t0), flush=True)
        else:
            with open(os.path.join(tmp, "err.txt"), "wb") as ef:
                r = subprocess.run([exe, fa, "-w", "10", "-o", os.path.join(tmp, "o.meme")], stdout=subprocess.DEVNULL, stderr=ef,
                                   env=dict(os.environ, PENGK_TIMING="1"))
            print("stderr to a file: gone %.3f" % (time.perf_counter() - t0), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
