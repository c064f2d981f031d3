/*
 * gasal_header.h -- the GASAL2 / AGAThA host API, re-implemented on libagatha_amd (MI355X).
 *
 * Source-level drop-in for clients written against the reference's AGAThA/src/gasal_header.h
 * (which pulls in gasal.h, args_parser.h, gasal_align.h, host_batch.h, ctors.h, interfaces.h):
 * same function names, argument meaning, public struct fields, result convention and error
 * behaviour ("[GASAL ERROR:] ..." on stderr + exit(EXIT_FAILURE), reference gasal.h:14-21).
 * Everything below the function boundary is new: no CUDA, no 0.98 GB scratch strip per stream,
 * device work goes through the C-ABI in agatha_amd.h.  One header instead of eight; the other
 * reference header names (gasal.h, args_parser.h, ...) are provided as forwarding stubs.
 *
 * Reference citations are AGAThA/src/<file>:<line>.
 */
#ifndef AGATHA_AMD_GASAL_HEADER_H
#define AGATHA_AMD_GASAL_HEADER_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <iostream>
#include <string>

#ifndef N_CODE
#define N_CODE 0x4E          /* padding base 'N' (reference AGAThA/Makefile:4) */
#endif

/* which of the two batches a call addresses (gasal.h:48-53) */
enum data_source { NONE, QUERY, TARGET, BOTH };

/* per-sequence operation codes taken from the FASTA header character (gasal.h:66-71, test_prog.cpp:83-92) */
enum operation_on_seq { FORWARD_NATURAL, REVERSE_NATURAL, FORWARD_COMPLEMENT, REVERSE_COMPLEMENT };

/* one pinned host page of an extensible batch (gasal.h:74-82) */
struct host_batch {
    uint8_t* data;
    uint32_t page_size;
    uint32_t data_size;
    uint32_t offset;
    int is_locked;
    struct host_batch* next;
};
typedef struct host_batch host_batch_t;

/* result arrays (gasal.h:85-94).  The reference leaves the start members NULL (res.cpp:27-28); here they are allocated and
 * filled when params->start_pos is set (extension, CLI flag -S: GASAL2's WITH_START, gasal.h:36), and the cigar members
 * when params->traceback is set (extension, CLI flag -T): pair k's path is n_cigar_ops[k] bytes at
 * cigar + host_query_batch_offsets[k] + host_target_batch_offsets[k], byte = (count << 2) | op, op 0 match / 1 mismatch /
 * 2 D / 3 I (include/agatha_amd.h: agatha_amd_align_traceback); n_cigar_ops[k] = 0xFFFFFFFF: no path */
struct gasal_res {
    int32_t* aln_score;
    int32_t* query_batch_end;
    int32_t* target_batch_end;
    int32_t* query_batch_start;
    int32_t* target_batch_start;
    uint8_t* cigar;
    uint32_t* n_cigar_ops;
};
typedef struct gasal_res gasal_res_t;

/* scoring + band parameters (gasal.h:165-173); identical layout to agatha_amd_scores */
typedef struct {
    int32_t match;
    int32_t mismatch;
    int32_t gap_open;
    int32_t gap_extend;
    int32_t slice_width;
    int32_t z_threshold;
    int32_t band_width;
} gasal_subst_scores;

/* per-stream state (gasal.h:97-155).  Public members keep the reference's names; device pointers are HIP
 * device pointers; `str` is a hipStream_t held as void* so that clients need no HIP headers. */
typedef struct {
    uint8_t* unpacked_query_batch;      /* device */
    uint8_t* unpacked_target_batch;     /* device */
    uint32_t* packed_query_batch;       /* device */
    uint32_t* packed_target_batch;      /* device */
    uint32_t* query_batch_offsets;      /* device */
    uint32_t* target_batch_offsets;     /* device */
    uint32_t* query_batch_lens;         /* device */
    uint32_t* target_batch_lens;        /* device */

    host_batch_t* extensible_host_unpacked_query_batch;
    host_batch_t* extensible_host_unpacked_target_batch;

    uint8_t* host_query_op;
    uint8_t* host_target_op;
    uint8_t* query_op;                  /* device */
    uint8_t* target_op;                 /* device */

    uint32_t* host_query_batch_offsets; /* pinned, filled by the caller */
    uint32_t* host_target_batch_offsets;
    uint32_t* host_query_batch_lens;
    uint32_t* host_target_batch_lens;

    gasal_res_t* host_res;              /* pinned host arrays: read these after gasal_is_aln_async_done()==0 */
    gasal_res_t* device_cpy;            /* host struct of device arrays */
    gasal_res_t* device_res;            /* kept for layout; equals device_cpy here (no device-side struct needed) */
    gasal_res_t* host_res_second;
    gasal_res_t* device_res_second;
    gasal_res_t* device_cpy_second;

    uint32_t gpu_max_query_batch_bytes;
    uint32_t gpu_max_target_batch_bytes;
    uint32_t host_max_query_batch_bytes;
    uint32_t host_max_target_batch_bytes;
    uint32_t gpu_max_n_alns;
    uint32_t host_max_n_alns;
    uint32_t current_n_alns;

    int32_t slice_width;
    uint32_t maximum_sequence_length;
    void* workspace;                    /* device scratch of agatha_amd_align (replaces global_buffer/host_buffer) */
    size_t workspace_bytes;

    void* str;                          /* hipStream_t */
    void* ev_begin;                     /* hipEvent_t pair for the -p timing mode */
    void* ev_end;
    int timing_pending;
    uint32_t timing_n_alns;                      /* pairs of the batch whose kernel time is pending (-p) */
    void* starts_scratch;               /* device scratch of agatha_amd_align_starts (only with params->start_pos) */
    size_t starts_scratch_bytes;
    void* tb_scratch;                   /* device scratch of agatha_amd_align_traceback (only with params->traceback) */
    size_t tb_scratch_bytes;
    size_t cigar_bytes;                 /* capacity of host_res->cigar / device_cpy->cigar */
    uint32_t cigar_alns;                /* capacity of host_res->n_cigar_ops / device_cpy->n_cigar_ops */
    void* timing_params;                /* Parameters* of the batch in flight: its raw_file gets the -p line */
    unsigned int* guard_host;           /* pinned: the int16 kernel's four guard counters of the batch in flight (agatha_amd_guard_stats), looked at when the batch is done */
    int is_free;
    int id;
} gasal_gpu_storage_t;

typedef struct {
    int n;
    gasal_gpu_storage_t* a;
} gasal_gpu_storage_v;

/* ---- command line (args_parser.h:15-68) ---- */
enum fail_type { NOT_ENOUGH_ARGS, TOO_MANY_ARGS, WRONG_ARG, WRONG_FILES, WRONG_ALGO };

class Parameters {
  public:
    Parameters(int argc, char** argv);
    ~Parameters();
    void print();
    void failure(fail_type f);
    void help();
    void parse();
    void fileopen();

    int32_t sa, sb, gapo, gape;
    int print_out;
    int n_threads;
    int slice_width, z_threshold, band_width;
    int32_t kernel_block_num, kernel_thread_num, kernel_align_num;
    bool isPacked;
    bool isReverseComplement;
    int start_pos;                      /* extension (-S): also compute query_batch_start / target_batch_start (WITH_START, gasal.h:36) */
    int traceback;                      /* extension (-T): also compute the alignment paths (cigar / n_cigar_ops, gasal.h:91-92) */
    int n_gpus;                         /* extension (-g): host threads are spread over this many GPUs (gasal_set_device) */
    bool isPacked2;                     /* extension (-K): host batches in the 2-bit + N-mask format (gasal_host_batch_fill_packed2): 3 bits per base over PCIe */
    std::string query_batch_fasta_filename, target_batch_fasta_filename, raw_filename;
    std::ifstream query_batch_fasta, target_batch_fasta;
    std::ofstream raw_file;

  private:
    int argc;
    char** argv;
};

/* ---- alignment (gasal_align.h:4-10) ---- */
void gasal_copy_subst_scores(gasal_subst_scores* subst);
void gasal_aln_async(gasal_gpu_storage_t* gpu_storage, const uint32_t actual_query_batch_bytes,
                     const uint32_t actual_target_batch_bytes, const uint32_t actual_n_alns, Parameters* params);
int gasal_is_aln_async_done(gasal_gpu_storage_t* gpu_storage);

/* ---- construction / destruction (ctors.h:5-15) ---- */
gasal_gpu_storage_v gasal_init_gpu_storage_v(int n_streams);
void gasal_init_streams(gasal_gpu_storage_v* gpu_storage_vec, int max_query_len, int max_target_len,
                        int32_t maximum_sequence_length, Parameters* params);
void gasal_destroy_streams(gasal_gpu_storage_v* gpu_storage_vec, Parameters* params);
void gasal_destroy_gpu_storage_v(gasal_gpu_storage_v* gpu_storage_vec);

/* ---- extensible host batches (host_batch.h:9-17) ---- */
host_batch_t* gasal_host_batch_new(uint32_t batch_bytes, uint32_t offset);
void gasal_host_batch_destroy(host_batch_t* res);
host_batch_t* gasal_host_batch_getlast(host_batch_t* arg);
void gasal_host_batch_reset(gasal_gpu_storage_t* gpu_storage);
uint32_t gasal_host_batch_fill(gasal_gpu_storage_t* gpu_storage, uint32_t idx, const char* data, uint32_t size, data_source SRC);
/* extension, for storages created with params->isPacked (ctors.cpp:65-73): packs `size` ASCII bases ON THE HOST (4 bit per
 * base, padded with N to a whole word) and appends the words to the batch; idx and the return value are offsets in the
 * UNPACKED layout (multiples of 8, what host_*_batch_offsets and gasal_aln_async's byte counts take), the page itself
 * holds idx / 2 bytes.  The H2D copy of such a batch is half the size and no pack kernel runs (gasal_align.cu:174). */
uint32_t gasal_host_batch_fill_packed(gasal_gpu_storage_t* gpu_storage, uint32_t idx, const char* data, uint32_t size, data_source SRC);
/* extension, for params->isPacked2: the same for the 2-bit + N-mask format (a uint16 of codes and a mask byte per eight bases; A, C,
 * G, T, N only -- any other letter is refused); the page holds idx / 8 * 3 bytes, gasal_aln_async ships them and expands them on the
 * device (agatha_amd_unpack2) instead of running the pack kernel. */
uint32_t gasal_host_batch_fill_packed2(gasal_gpu_storage_t* gpu_storage, uint32_t idx, const char* data, uint32_t size, data_source SRC);
uint32_t gasal_host_batch_add(gasal_gpu_storage_t* gpu_storage, uint32_t idx, const char* data, uint32_t size, data_source SRC);
uint32_t gasal_host_batch_addbase(gasal_gpu_storage_t* gpu_storage, uint32_t idx, const char base, data_source SRC);
void gasal_host_batch_print(host_batch_t* res);
void gasal_host_batch_printall(host_batch_t* res);

/* ---- results (res.h:4-9) ---- */
gasal_res_t* gasal_res_new_host(uint32_t max_n_alns, Parameters* params);
gasal_res_t* gasal_res_new_device(gasal_res_t* device_cpy);
gasal_res_t* gasal_res_new_device_cpy(uint32_t max_n_alns, Parameters* params);
void gasal_res_destroy_host(gasal_res_t* res);
void gasal_res_destroy_device(gasal_res_t* device_res, gasal_res_t* device_cpy);

/* ---- misc (interfaces.h:9-14) ---- */
void gasal_host_alns_resize(gasal_gpu_storage_t* gpu_storage, int new_max_alns, Parameters* params);
void gasal_op_fill(gasal_gpu_storage_t* gpu_storage_t, uint8_t* data, uint32_t nbr_seqs_in_stream, data_source SRC);
void gasal_set_device(int gpu_select = 0, bool isPrintingProp = true);

#endif /* AGATHA_AMD_GASAL_HEADER_H */
