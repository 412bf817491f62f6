import sys, time, os
sys.path.insert(0,"tests")
import bwalib as B, ctypes
seqs=B.synth_genome(3000000, n_contigs=6, seed=11, repeat_frac=0.08, n_frac=0.0005)
B.write_fasta("/tmp/m.fa",seqs)
lib=B.product_lib(); build=lib.dll.jnibwa_createReferenceIndex; build.argtypes=[ctypes.c_char_p]*3
assert build(b"/tmp/m.fa",b"/tmp/m.fa",b"auto")==0
assert lib.create_index_file("/tmp/m.fa","/tmp/m.img")==0
rd=[b'GGTGGTTTGGCGCTAATATGACTGTTTGGACACTAATTGTCTGCTCTAACCAGATTACCTTGGTTAACGTCATGCCTCACAAGGTGGCATNGATAGCATCCTATCCGGCACCCACTTAGTCCGTGGGCTTCCCGGTGGGACAGATACTTA']
h=lib.open_index("/tmp/m.img")
a=lib.align_raw(h,lib.default_options(),B.pack_request(rd)); print("gpu", len(a) if a else None, flush=True)
