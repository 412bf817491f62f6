import sys, time, os
sys.path.insert(0,"tests")
import bwalib as B, ctypes
n=int(sys.argv[1])
seqs=B.synth_genome(3000000, n_contigs=6, seed=11, repeat_frac=0.08, n_frac=0.0005)
B.write_fasta("/tmp/m.fa",seqs)
lib=B.product_lib(); build=lib.dll.jnibwa_createReferenceIndex; build.argtypes=[ctypes.c_char_p]*3
assert build(b"/tmp/m.fa",b"/tmp/m.fa",b"auto")==0
assert lib.create_index_file("/tmp/m.fa","/tmp/m.img")==0
reads=B.simulate_reads(seqs,n,length=150,seed=42)
h=lib.open_index("/tmp/m.img")
t=time.time(); a=lib.align_raw(h,lib.default_options(),B.pack_request(reads)); print("gpu",time.time()-t, len(a) if a else None, flush=True)
B.build_oracle(); orc=B.oracle_lib(); ho=orc.open_index("/tmp/m.img")
t=time.time(); b=orc.align_raw(ho,orc.default_options(),B.pack_request(reads)); print("oracle",time.time()-t, a==b, flush=True)
