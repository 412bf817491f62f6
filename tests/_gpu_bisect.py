import sys, time, os, subprocess, pickle
sys.path.insert(0, "tests")
import bwalib as B, ctypes
if len(sys.argv) > 1 and sys.argv[1] == "child":
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    reads = pickle.load(open("/tmp/reads.pkl", "rb"))[lo:hi]
    lib = B.product_lib(); h = lib.open_index("/tmp/m.img")
    a = lib.align_raw(h, lib.default_options(), B.pack_request(reads))
    print("done", len(a) if a else None); sys.exit(0)
seqs = B.synth_genome(3000000, n_contigs=6, seed=11, repeat_frac=0.08, n_frac=0.0005)
B.write_fasta("/tmp/m.fa", seqs)
lib = B.product_lib(); build = lib.dll.jnibwa_createReferenceIndex; build.argtypes = [ctypes.c_char_p] * 3
assert build(b"/tmp/m.fa", b"/tmp/m.fa", b"auto") == 0
assert lib.create_index_file("/tmp/m.fa", "/tmp/m.img") == 0
reads = B.simulate_reads(seqs, 20000, length=150, seed=42)
pickle.dump(reads, open("/tmp/reads.pkl", "wb"))
def bad(lo, hi):
    try:
        r = subprocess.run([sys.executable, "tests/_gpu_bisect.py", "child", str(lo), str(hi)], timeout=25, capture_output=True, text=True)
        print("  range", lo, hi, "rc", r.returncode, r.stdout.strip()[-40:], r.stderr.strip()[-200:], flush=True)
        return r.returncode != 0
    except subprocess.TimeoutExpired:
        print("  range", lo, hi, "TIMEOUT", flush=True)
        return True
lo, hi = 0, len(reads)
assert bad(lo, hi)
while hi - lo > 1:
    mid = (lo + hi) // 2
    if bad(lo, mid): hi = mid
    elif bad(mid, hi): lo = mid
    else:
        print("neither half fails alone", lo, mid, hi); break
print("CULPRIT", lo, hi, reads[lo:hi], flush=True)
