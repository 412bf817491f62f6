// tests/gpu_units/drive.cpp -- minimal torch-free driver for profiling runs (rocprofv3 --pmc is not
// reliable around a Python/torch process): open an index image, upload a request file, align it n times.
#include <stdio.h>
#include <stdlib.h>
#include <fcntl.h>
#include <vector>
#include "../../include/bwamem_hip.h"

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: drive <image> <request-file> [iterations] [extra-flag-bits]\n"); return 2; }
    int iters = argc > 3 ? atoi(argv[3]) : 1;
    int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror(argv[1]); return 1; }
    bwaidx_t* idx = jnibwa_openIndex(fd);
    if (!idx) { fprintf(stderr, "openIndex failed\n"); return 1; }
    FILE* fp = fopen(argv[2], "rb");
    if (!fp) { perror(argv[2]); return 1; }
    fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
    std::vector<char> req((size_t)n);
    if (fread(req.data(), 1, (size_t)n, fp) != (size_t)n) return 1;
    fclose(fp);
    bwamem_batch_t* b = bwamem_hip_batch_upload(idx, req.data(), (size_t)n);
    if (!b) return 1;
    mem_opt_t* opt = jnibwa_createDefaultOptions();
    if (argc > 4) *(int*)((char*)opt + 60) |= (int)strtol(argv[4], 0, 0);      // extra mem_opt_t.flag bits (0x2 = paired-end)
    bwamem_hip_stats_enable(1);
    for (int i = 0; i < iters; ++i)
        if (bwamem_hip_batch_align(idx, opt, 0, b, 0) != 0) { fprintf(stderr, "align failed\n"); return 1; }
    bwamem_stats_t st; bwamem_hip_stats_get(&st);
    printf("reads=%llu n_ext=%llu ms_seed=%.2f ms_extend=%.2f ms_final=%.2f ms_sa=%.2f result_bytes=%zu\n", (unsigned long long)st.n_reads,
           (unsigned long long)st.n_ext, st.ms_seed, st.ms_extend, st.ms_final, st.ms_sa, bwamem_hip_batch_result_bytes(b));
    bwamem_hip_batch_free(b);
    jnibwa_free(opt);
    jnibwa_destroyIndex(idx);
    return 0;
}
