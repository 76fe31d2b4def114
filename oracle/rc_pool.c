/*
 * rc_pool.c -- block-parallel timing harness for the CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see refcpu.h).
 *
 * the reference decodes one block per call on one thread; "all host cores" therefore means one
 * independent unit per task.  Every worker thread cycles over its share of the units (k, k + threads, ...) and calls
 * the oracle's ordinary single-unit entry point until the time budget is used.  Used by bench.py's cpu_context leg.
 */
#define _POSIX_C_SOURCE 200809L
#include <malloc.h>
#include <pthread.h>
#include <stdlib.h>
#include <time.h>
#include "refcpu.h"

typedef struct {
    int codec, aux, threads, index, failed;
    const uint8_t* const* ins;
    const size_t* lens;
    size_t n;
    double deadline;
    uint64_t out_bytes, in_bytes, units;
} pool_arg;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int decode_one(int codec, int aux, const uint8_t* in, size_t len, size_t* produced) {
    uint8_t* out = NULL;
    size_t n = 0, used = 0;
    int st;
    switch (codec) {
        case 1: st = refcpu_deflate_decompress(in, len, &out, &n, &used); break;
        case 2: st = refcpu_lz4_block(in, len, NULL, 0, &out, &n); break;
        case 3: st = refcpu_bzip2_decompress(in, len, &out, &n, &used); break;
        case 4: st = refcpu_lzma2_decompress(in, len, (uint8_t)aux, &out, &n, &used); break;
        default: st = SWC_E_INVALID_ARGUMENT;
    }
    refcpu_free(out);
    *produced = n;
    return st;
}

static void* worker(void* p) {
    pool_arg* a = (pool_arg*)p;
    size_t i = (size_t)a->index % a->n;
    while (now_s() < a->deadline) {
        size_t n = 0;
        if (decode_one(a->codec, a->aux, a->ins[i], a->lens[i], &n) != SWC_OK) { a->failed = 1; break; }
        a->out_bytes += n;
        a->in_bytes += a->lens[i];
        a->units += 1;
        i = (i + (size_t)a->threads) % a->n;
    }
    return NULL;
}

/* codec: 1 raw Deflate, 2 LZ4 block, 3 bzip2 stream, 4 raw LZMA2 (aux = dictionary-size byte).  Returns the elapsed
 * seconds (< 0 on a decode failure); totals over all threads in *out_bytes / *in_bytes / *units. */
double refcpu_timed_pool(int codec, int aux, const uint8_t* const* ins, const size_t* lens, size_t n, int threads,
                         double seconds, uint64_t* out_bytes, uint64_t* in_bytes, uint64_t* units) {
    if (n == 0 || threads < 1) return -1.0;
    /* The restatement allocates like the reference does -- three decoding trees of up to 2^16 Ints per Deflate block, an
     * output array that grows by doubling -- and with glibc's defaults every such block above 128 KiB is its own mmap /
     * munmap: 256 threads then queue on the process's address-space lock, not on the decode (round 2: 12x on 256
     * threads).  Keep freed memory in per-thread arenas instead. */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_ARENA_MAX, threads + 1);
    size_t warm = 0;
    if (decode_one(codec, aux, ins[0], lens[0], &warm) != SWC_OK) return -1.0;   /* also builds the lazy static tables */
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    pool_arg* args = (pool_arg*)calloc((size_t)threads, sizeof(pool_arg));
    const double t0 = now_s();
    for (int k = 0; k < threads; k++) {
        pool_arg a = { codec, aux, threads, k, 0, ins, lens, n, t0 + seconds, 0, 0, 0 };
        args[k] = a;
        pthread_create(&th[k], NULL, worker, &args[k]);
    }
    int failed = 0;
    *out_bytes = *in_bytes = *units = 0;
    for (int k = 0; k < threads; k++) {
        pthread_join(th[k], NULL);
        failed |= args[k].failed;
        *out_bytes += args[k].out_bytes;
        *in_bytes += args[k].in_bytes;
        *units += args[k].units;
    }
    const double dt = now_s() - t0;
    free(th);
    free(args);
    return failed ? -1.0 : dt;
}
