// gasal_host.cpp -- the GASAL2/AGAThA host API (include/gasal_header.h) implemented over the C-ABI of
// libagatha_amd.so.  Plain C++ (no HIP headers): every device operation is an agatha_amd_* call.
//
// What each function replaces in the reference (AGAThA/src/...):
//   gasal_copy_subst_scores        gasal_align.cu:295-309  (8 cudaMemcpyToSymbol -> one process-global struct)
//   gasal_init_gpu_storage_v/..    ctors.cpp:17-173
//   gasal_host_batch_*             host_batch.cpp:11-246   (defines the batch wire format)
//   gasal_res_*                    res.cpp:8-119
//   gasal_host_alns_resize/op_fill/set_device   interfaces.cpp:26-116
//   gasal_aln_async / gasal_is_aln_async_done   gasal_align.cu:27-292
#include "../../include/gasal_header.h"
#include "../../include/agatha_amd.h"

#include <algorithm>
#include <mutex>

namespace {

// Process-global like the reference's __constant__ symbols (gasal_kernels.h:29-36): set it once, before any thread
// aligns; gasal_aln_async reads it unsynchronised.
gasal_subst_scores g_scores = {2, 4, 4, 2, 3, 400, 751};    // defaults of args_parser.cpp:12-22
std::mutex g_raw_mutex;                                     // serialises the -p lines of concurrent host threads

[[noreturn]] void die_hip(int rc, int line)
{
    fprintf(stderr, "[GASAL HIP ERROR:] %s: %s. Line no. %d in file %s\n", agatha_amd_strerror(rc),
            agatha_amd_last_error(), line, __FILE__);
    exit(EXIT_FAILURE);
}
#define CHK(call) do { int rc_ = (call); if (rc_ != 0) die_hip(rc_, __LINE__); } while (0)

template <typename T> T* dev_alloc(size_t n) { void* p = nullptr; CHK(agatha_amd_malloc(&p, n * sizeof(T))); return (T*)p; }
template <typename T> T* pin_alloc(size_t n) { void* p = nullptr; CHK(agatha_amd_host_alloc(&p, n * sizeof(T))); return (T*)p; }
void dev_free(void* p) { if (p) CHK(agatha_amd_free(p)); }
void pin_free(void* p) { if (p) CHK(agatha_amd_host_free(p)); }

template <typename T> T* pin_realloc(T* src, int new_n, int old_n)
{
    if (new_n < old_n) {
        fprintf(stderr, "[GASAL ERROR] host realloc: invalid sizes. New size < old size (%d < %d)", new_n, old_n);
        exit(EXIT_FAILURE);
    }
    T* dst = pin_alloc<T>((size_t)new_n);
    memcpy(dst, src, (size_t)old_n * sizeof(T));
    pin_free(src);
    return dst;
}

uint32_t pad8(uint32_t v) { return (v + 7u) & ~7u; }

void alloc_device_meta(gasal_gpu_storage_t* s, uint32_t n, uint32_t max_query_len = 0, uint32_t max_target_len = 0)
{
    s->query_batch_lens = dev_alloc<uint32_t>(n);
    s->target_batch_lens = dev_alloc<uint32_t>(n);
    s->query_batch_offsets = dev_alloc<uint32_t>(n);
    s->target_batch_offsets = dev_alloc<uint32_t>(n);
    s->query_op = dev_alloc<uint8_t>(n);
    s->target_op = dev_alloc<uint8_t>(n);
    s->workspace_bytes = agatha_amd_workspace_bytes_long(n, max_query_len, max_target_len);
    s->workspace = dev_alloc<uint8_t>(s->workspace_bytes);
}
void free_device_meta(gasal_gpu_storage_t* s)
{
    dev_free(s->query_batch_lens); dev_free(s->target_batch_lens);
    dev_free(s->query_batch_offsets); dev_free(s->target_batch_offsets);
    dev_free(s->query_op); dev_free(s->target_op); dev_free(s->workspace);
    s->query_batch_lens = s->target_batch_lens = s->query_batch_offsets = s->target_batch_offsets = nullptr;
    s->query_op = s->target_op = nullptr; s->workspace = nullptr;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ scores
void gasal_copy_subst_scores(gasal_subst_scores* subst) { g_scores = *subst; }

// ------------------------------------------------------------------------------------------------ results
// (the start members: NULL as in the reference, res.cpp:27-28,76-77, unless params->start_pos asks for them)
gasal_res_t* gasal_res_new_host(uint32_t max_n_alns, Parameters* params)
{
    gasal_res_t* r = (gasal_res_t*)calloc(1, sizeof(gasal_res_t));
    if (!r) { fprintf(stderr, "Malloc error on res host "); exit(1); }
    r->aln_score = pin_alloc<int32_t>(max_n_alns);
    r->query_batch_end = pin_alloc<int32_t>(max_n_alns);
    r->target_batch_end = pin_alloc<int32_t>(max_n_alns);
    if (params && params->start_pos) {
        r->query_batch_start = pin_alloc<int32_t>(max_n_alns);
        r->target_batch_start = pin_alloc<int32_t>(max_n_alns);
    }
    return r;
}
gasal_res_t* gasal_res_new_device_cpy(uint32_t max_n_alns, Parameters* params)
{
    gasal_res_t* r = (gasal_res_t*)calloc(1, sizeof(gasal_res_t));
    r->aln_score = dev_alloc<int32_t>(max_n_alns);
    r->query_batch_end = dev_alloc<int32_t>(max_n_alns);
    r->target_batch_end = dev_alloc<int32_t>(max_n_alns);
    if (params && params->start_pos) {
        r->query_batch_start = dev_alloc<int32_t>(max_n_alns);
        r->target_batch_start = dev_alloc<int32_t>(max_n_alns);
    }
    return r;
}
// The reference keeps a second, device-resident copy of the struct for its kernel to dereference; the HIP kernel
// takes the three arrays directly, so the "device struct" is the host-side struct of device pointers itself.
gasal_res_t* gasal_res_new_device(gasal_res_t* device_cpy) { return device_cpy; }
void gasal_res_destroy_host(gasal_res_t* r)
{
    if (!r) return;
    pin_free(r->aln_score); pin_free(r->query_batch_start); pin_free(r->target_batch_start);
    pin_free(r->query_batch_end); pin_free(r->target_batch_end);
    pin_free(r->cigar); pin_free(r->n_cigar_ops);
    free(r);
}
void gasal_res_destroy_device(gasal_res_t* device_res, gasal_res_t* device_cpy)
{
    (void)device_res;
    if (!device_cpy) return;
    dev_free(device_cpy->aln_score); dev_free(device_cpy->query_batch_start); dev_free(device_cpy->target_batch_start);
    dev_free(device_cpy->query_batch_end); dev_free(device_cpy->target_batch_end);
    dev_free(device_cpy->cigar); dev_free(device_cpy->n_cigar_ops);
    free(device_cpy);
}

// ------------------------------------------------------------------------------------------------ host batches
host_batch_t* gasal_host_batch_new(uint32_t batch_bytes, uint32_t offset)
{
    host_batch_t* p = (host_batch_t*)calloc(1, sizeof(host_batch_t));
    p->data = pin_alloc<uint8_t>(batch_bytes);
    p->page_size = batch_bytes;
    p->offset = offset;
    return p;
}
void gasal_host_batch_destroy(host_batch_t* p)
{
    if (!p) { fprintf(stderr, "[GASAL ERROR] Trying to free a NULL pointer\n"); exit(1); }
    while (p) { host_batch_t* nx = p->next; pin_free(p->data); free(p); p = nx; }
}
host_batch_t* gasal_host_batch_getlast(host_batch_t* p) { while (p->next) p = p->next; return p; }

void gasal_host_batch_reset(gasal_gpu_storage_t* s)
{
    host_batch_t* heads[2] = {s->extensible_host_unpacked_query_batch, s->extensible_host_unpacked_target_batch};
    for (host_batch_t* p : heads)
        for (; p; p = p->next) { p->data_size = 0; p->offset = 0; p->is_locked = 0; }
}

static void pick_side(gasal_gpu_storage_t* s, data_source src, host_batch_t** page, uint32_t** total)
{
    if (src == QUERY) { *page = s->extensible_host_unpacked_query_batch; *total = &s->host_max_query_batch_bytes; }
    else if (src == TARGET) { *page = s->extensible_host_unpacked_target_batch; *total = &s->host_max_target_batch_bytes; }
    else { fprintf(stderr, "[GASAL ERROR:] host batch call needs QUERY or TARGET\n"); exit(EXIT_FAILURE); }
}

// Appends `size` bases at batch offset idx, pads with 'N' to a multiple of 8, returns the new offset.  Pages are
// filled in order; a full page is locked and a new page of twice its size is chained (host_batch.cpp:79-154).
uint32_t gasal_host_batch_fill(gasal_gpu_storage_t* s, uint32_t idx, const char* data, uint32_t size, data_source SRC)
{
    host_batch_t* page; uint32_t* total;
    pick_side(s, SRC, &page, &total);
    const uint32_t need = pad8(size);
    while (page->is_locked) page = page->next;
    if (page->page_size - page->data_size < need) {
        if (!page->next) {
            uint32_t sz = page->page_size * 2;
            while (sz < need) sz *= 2;
            page->next = gasal_host_batch_new(sz, page->offset + page->data_size);
            *total += sz;
        } else {
            page->next->offset = page->offset + page->data_size;
        }
        page->is_locked = 1;
        page = page->next;
    }
    if (page->page_size - page->data_size >= need) {
        uint8_t* dst = page->data + (idx - page->offset);
        memcpy(dst, data, size);
        memset(dst + size, N_CODE, need - size);
        page->data_size += need;
        idx += need;
    }
    return idx;
}

// Host-side packing for isPacked storages (extension; see include/gasal_header.h).  Same page discipline as
// gasal_host_batch_fill, in packed bytes: a sequence of `size` bases takes pad8(size) / 2 bytes.
uint32_t gasal_host_batch_fill_packed(gasal_gpu_storage_t* s, uint32_t idx, const char* data, uint32_t size, data_source SRC)
{
    host_batch_t* page; uint32_t* total;
    pick_side(s, SRC, &page, &total);
    const uint32_t padded = pad8(size), need = padded / 2, at = idx / 2;
    while (page->is_locked) page = page->next;
    if (page->page_size - page->data_size < need) {
        if (!page->next) {
            uint32_t sz = page->page_size * 2;
            while (sz < need) sz *= 2;
            page->next = gasal_host_batch_new(sz, page->offset + page->data_size);
            *total += sz;
        } else {
            page->next->offset = page->offset + page->data_size;
        }
        page->is_locked = 1;
        page = page->next;
    }
    uint8_t* dst = page->data + (at - page->offset);
    const uint32_t whole = size & ~7u;
    CHK(agatha_amd_pack_host((const uint8_t*)data, whole, (uint32_t*)dst));
    if (padded != whole) {
        uint8_t tail[8];
        memset(tail, N_CODE, 8);
        memcpy(tail, data + whole, size - whole);
        CHK(agatha_amd_pack_host(tail, 8, (uint32_t*)(dst + whole / 2)));
    }
    page->data_size += need;
    return idx + padded;
}

// Host-side packing into the 2-bit + N-mask format (extension, params->isPacked2; the "2-bit+N-mask" option of SURVEY.md 8 f3):
// 3 bits per base over PCIe.  A page of page_size bytes holds page_size / 3 words of eight bases: their uint16 codes in its
// first two thirds, their mask bytes in the last third; offset and data_size count 3 bytes per word.  Letters other than
// A, C, G, T, N (any case) have no code in this format: the call refuses them (the 4-bit formats keep them for the compare kernel).
uint32_t gasal_host_batch_fill_packed2(gasal_gpu_storage_t* s, uint32_t idx, const char* data, uint32_t size, data_source SRC)
{
    host_batch_t* page; uint32_t* total;
    pick_side(s, SRC, &page, &total);
    const uint32_t padded = pad8(size), nw = padded / 8, need = 3 * nw, at = idx / 8 * 3;
    while (page->is_locked) page = page->next;
    if (page->page_size / 3 * 3 - page->data_size < need) {
        if (!page->next) {
            uint32_t sz = page->page_size * 2;
            while (sz / 3 * 3 < need) sz *= 2;
            page->next = gasal_host_batch_new(sz, page->offset + page->data_size);
            *total += sz;
        } else {
            page->next->offset = page->offset + page->data_size;
        }
        page->is_locked = 1;
        page = page->next;
    }
    const uint32_t cap_words = page->page_size / 3, wi = (at - page->offset) / 3;
    uint16_t* codes = (uint16_t*)page->data + wi;
    uint8_t* mask = page->data + 2 * (size_t)cap_words + wi;
    const uint32_t whole = size & ~7u;
    long other = agatha_amd_pack2_host((const uint8_t*)data, whole, codes, mask);
    if (other >= 0 && padded != whole) {
        uint8_t tail[8];
        memset(tail, 'N', 8);
        memcpy(tail, data + whole, size - whole);
        const long o2 = agatha_amd_pack2_host(tail, 8, codes + whole / 8, mask + whole / 8);
        other = o2 < 0 ? o2 : other + o2;
    }
    if (other < 0) CHK((int)other);
    if (other > 0) {
        fprintf(stderr, "[GASAL ERROR:] the 2-bit format (isPacked2) holds A, C, G, T and N only: %ld other letters in a sequence\n", other);
        exit(EXIT_FAILURE);
    }
    page->data_size += need;
    return idx + padded;
}

// Raw append without padding (host_batch.cpp:157-225)
uint32_t gasal_host_batch_add(gasal_gpu_storage_t* s, uint32_t idx, const char* data, uint32_t size, data_source SRC)
{
    host_batch_t* page; uint32_t* total;
    pick_side(s, SRC, &page, &total);
    for (;;) {
        const bool fits_total = *total >= idx + size;
        if (fits_total && (!page->next || page->next->offset >= idx + size)) {
            memcpy(page->data + (idx - page->offset), data, size);
            if (idx + size - page->offset > page->data_size) page->data_size = idx + size - page->offset;
            return idx + size;
        }
        if (fits_total && page->next) { page = page->next; continue; }
        uint32_t grow = *total;
        while (grow < size) grow += grow;
        host_batch_t* last = gasal_host_batch_getlast(page);
        last->next = gasal_host_batch_new(grow, idx);
        *total += grow;
        page = last->next;
    }
}
uint32_t gasal_host_batch_addbase(gasal_gpu_storage_t* s, uint32_t idx, const char base, data_source SRC)
{
    return gasal_host_batch_add(s, idx, &base, 1, SRC);
}
void gasal_host_batch_print(host_batch_t* p)
{
    fprintf(stderr, "[GASAL PRINT] Page data: offset=%d, next_offset=%d, data size=%d, page size=%d\n", p->offset,
            (p->next ? (int)p->next->offset : -1), p->data_size, p->page_size);
}
void gasal_host_batch_printall(host_batch_t* p)
{
    for (bool first = true; p; p = p->next, first = false) { if (!first) fprintf(stderr, "+--->"); gasal_host_batch_print(p); }
}

// ------------------------------------------------------------------------------------------------ storage
gasal_gpu_storage_v gasal_init_gpu_storage_v(int n_streams)
{
    gasal_gpu_storage_v v;
    v.a = (gasal_gpu_storage_t*)calloc((size_t)n_streams, sizeof(gasal_gpu_storage_t));
    v.n = n_streams;
    return v;
}

void gasal_init_streams(gasal_gpu_storage_v* vec, int max_query_len, int max_target_len,
                        int32_t maximum_sequence_length, Parameters* params)
{
    const uint32_t n = (uint32_t)params->kernel_align_num;
    const uint64_t qbytes64 = (uint64_t)n * pad8((uint32_t)max_query_len);
    const uint64_t tbytes64 = (uint64_t)n * pad8((uint32_t)max_target_len);
    if (qbytes64 > 0xFFFFFFF0ull || tbytes64 > 0xFFFFFFF0ull) {
        fprintf(stderr, "[GASAL ERROR:] kernel_align_num x max length exceeds the 32-bit batch size of the GASAL API; lower -a\n");
        exit(EXIT_FAILURE);
    }
    const uint32_t qbytes = (uint32_t)qbytes64, tbytes = (uint32_t)tbytes64;
    if (params->isPacked && params->isPacked2) { fprintf(stderr, "[GASAL ERROR:] isPacked and isPacked2 are two formats of the host batches: choose one\n"); exit(EXIT_FAILURE); }
    for (int i = 0; i < vec->n; i++) {
        gasal_gpu_storage_t* s = &vec->a[i];
        s->extensible_host_unpacked_query_batch = gasal_host_batch_new(qbytes, 0);
        s->extensible_host_unpacked_target_batch = gasal_host_batch_new(tbytes, 0);
        s->unpacked_query_batch = dev_alloc<uint8_t>(qbytes + 16);
        s->unpacked_target_batch = dev_alloc<uint8_t>(tbytes + 16);
        if (params->isPacked) {          // caller ships packed words: ctors.cpp:65-73
            s->packed_query_batch = (uint32_t*)s->unpacked_query_batch;
            s->packed_target_batch = (uint32_t*)s->unpacked_target_batch;
        } else {
            s->packed_query_batch = dev_alloc<uint32_t>(qbytes / 8 + 4);
            s->packed_target_batch = dev_alloc<uint32_t>(tbytes / 8 + 4);
        }
        s->host_query_op = pin_alloc<uint8_t>(n);
        s->host_target_op = pin_alloc<uint8_t>(n);
        memset(s->host_query_op, 0, n); memset(s->host_target_op, 0, n);
        s->host_query_batch_lens = pin_alloc<uint32_t>(n);
        s->host_target_batch_lens = pin_alloc<uint32_t>(n);
        s->host_query_batch_offsets = pin_alloc<uint32_t>(n);
        s->host_target_batch_offsets = pin_alloc<uint32_t>(n);
        alloc_device_meta(s, n, (uint32_t)std::max(max_query_len, 1), (uint32_t)std::max(max_target_len, 1));
        s->host_res = gasal_res_new_host(n, params);
        s->device_cpy = gasal_res_new_device_cpy(n, params);
        s->device_res = gasal_res_new_device(s->device_cpy);
        CHK(agatha_amd_stream_create(&s->str));
        CHK(agatha_amd_event_create(&s->ev_begin));
        CHK(agatha_amd_event_create(&s->ev_end));
        s->guard_host = pin_alloc<unsigned int>(4);
        memset(s->guard_host, 0, 4 * sizeof(unsigned int));
        s->is_free = 1;
        s->host_max_query_batch_bytes = s->gpu_max_query_batch_bytes = qbytes;
        s->host_max_target_batch_bytes = s->gpu_max_target_batch_bytes = tbytes;
        s->host_max_n_alns = s->gpu_max_n_alns = n;
        s->current_n_alns = 0;
        s->slice_width = params->slice_width;
        s->maximum_sequence_length = (uint32_t)maximum_sequence_length;
        s->id = i;
    }
}

void gasal_destroy_streams(gasal_gpu_storage_v* vec, Parameters* params)
{
    for (int i = 0; i < vec->n; i++) {
        gasal_gpu_storage_t* s = &vec->a[i];
        if (s->str) CHK(agatha_amd_stream_synchronize(s->str));
        gasal_host_batch_destroy(s->extensible_host_unpacked_query_batch);
        gasal_host_batch_destroy(s->extensible_host_unpacked_target_batch);
        gasal_res_destroy_host(s->host_res);
        gasal_res_destroy_device(s->device_res, s->device_cpy);
        pin_free(s->host_query_op); pin_free(s->host_target_op);
        pin_free(s->host_query_batch_offsets); pin_free(s->host_target_batch_offsets);
        pin_free(s->host_query_batch_lens); pin_free(s->host_target_batch_lens);
        free_device_meta(s);
        dev_free(s->starts_scratch); dev_free(s->tb_scratch);
        dev_free(s->unpacked_query_batch); dev_free(s->unpacked_target_batch);
        if (!params->isPacked) { dev_free(s->packed_query_batch); dev_free(s->packed_target_batch); }
        pin_free(s->guard_host);
        if (s->ev_begin) CHK(agatha_amd_event_destroy(s->ev_begin));
        if (s->ev_end) CHK(agatha_amd_event_destroy(s->ev_end));
        if (s->str) CHK(agatha_amd_stream_destroy(s->str));
        memset(s, 0, sizeof(*s));
    }
}

void gasal_destroy_gpu_storage_v(gasal_gpu_storage_v* vec) { if (vec->a) free(vec->a); vec->a = nullptr; }

// ------------------------------------------------------------------------------------------------ misc
void gasal_host_alns_resize(gasal_gpu_storage_t* s, int new_max_alns, Parameters* params)
{
    fprintf(stderr, "[GASAL WARNING] Resizing gpu_storage from %d sequences to %d sequences... ", s->host_max_n_alns, new_max_alns);
    const int old = (int)s->host_max_n_alns;
    s->host_query_op = pin_realloc(s->host_query_op, new_max_alns, old);
    s->host_target_op = pin_realloc(s->host_target_op, new_max_alns, old);
    s->host_query_batch_lens = pin_realloc(s->host_query_batch_lens, new_max_alns, old);
    s->host_target_batch_lens = pin_realloc(s->host_target_batch_lens, new_max_alns, old);
    s->host_query_batch_offsets = pin_realloc(s->host_query_batch_offsets, new_max_alns, old);
    s->host_target_batch_offsets = pin_realloc(s->host_target_batch_offsets, new_max_alns, old);
    gasal_res_destroy_host(s->host_res);
    s->host_res = gasal_res_new_host((uint32_t)new_max_alns, params);
    s->host_max_n_alns = (uint32_t)new_max_alns;   // device side grows lazily in gasal_aln_async
    fprintf(stderr, " done. This can harm performance.\n");
}

void gasal_op_fill(gasal_gpu_storage_t* s, uint8_t* data, uint32_t nbr_seqs_in_stream, data_source SRC)
{
    uint8_t* dst = (SRC == QUERY) ? s->host_query_op : (SRC == TARGET ? s->host_target_op : nullptr);
    if (dst) memcpy(dst, data, nbr_seqs_in_stream);
}

void gasal_set_device(int gpu_select, bool isPrintingProp)
{
    const int n = agatha_amd_device_count();
    if (isPrintingProp) fprintf(stderr, "Found %d GPUs\n", n);
    if (gpu_select > n - 1) {
        fprintf(stderr, "Error: can't select device %d when only %d devices are selected (range from 0 to %d)\n", gpu_select, n, n - 1);
        exit(EXIT_FAILURE);
    }
    CHK(agatha_amd_set_device(gpu_select));
    if (isPrintingProp) fprintf(stderr, "Selected device %d\n", gpu_select);
}

// ------------------------------------------------------------------------------------------------ the hot path
// How many batches' align kernels may be on one GPU at once (round 6).  Every storage has its own stream, a client thread usually two storages
// (the reference's NB_STREAMS, test_prog.cpp:12), and every align call is a persistent grid that fills the chip: two of them overlap usefully --
// the tail of one under the head of the next, 1.06 x the single-batch rate in the steady state -- but with four client threads eight were
// queued at once, whichever workgroups got on the chip belonged to five or six different grids and lane groups waited for partners that were
// not resident.  So the align kernels of batch k of a device wait, on the device, for those of batch k - LIMIT (an event ring; the host never
// blocks, H2D copies and the host's filling of other storages go on): more client threads then only add host-side overlap.
// AGATHA_AMD_MAX_INFLIGHT: the limit (default 2, 0 = none).  Measured over 16-batch feeds (tools/pipe_sweep.py, profiles/r06_v1/pipe_sweep.txt): within the
// run-to-run spread (+- 5 %) of no limit for two threads; four threads stay behind two either way (what the kernel traces say about that).
namespace {
constexpr int kGateRing = 64, kGateDevices = 64;
struct DeviceGate { std::mutex mu; unsigned long long ticket = 0; void* done[kGateRing] = {nullptr}; };
DeviceGate g_gate[kGateDevices];
int gate_limit()
{
    static const int v = [] { const char* e = getenv("AGATHA_AMD_MAX_INFLIGHT"); const int x = (e && *e) ? atoi(e) : 2; return x < 0 ? 0 : (x > kGateRing / 2 ? kGateRing / 2 : x); }();
    return v;
}
}  // namespace

void gasal_aln_async(gasal_gpu_storage_t* s, const uint32_t actual_query_batch_bytes,
                     const uint32_t actual_target_batch_bytes, const uint32_t actual_n_alns, Parameters* params)
{
    // argument checks: same conditions and messages as gasal_align.cu:33-68
    if (actual_n_alns <= 0) { fprintf(stderr, "[GASAL ERROR:] actual_n_alns <= 0\n"); exit(EXIT_FAILURE); }
    if (actual_query_batch_bytes <= 0) { fprintf(stderr, "[GASAL ERROR:] actual_query_batch_bytes <= 0\n"); exit(EXIT_FAILURE); }
    if (actual_target_batch_bytes <= 0) { fprintf(stderr, "[GASAL ERROR:] actual_target_batch_bytes <= 0\n"); exit(EXIT_FAILURE); }
    if (actual_query_batch_bytes % 8) { fprintf(stderr, "[GASAL ERROR:] actual_query_batch_bytes=%d is not a multiple of 8\n", actual_query_batch_bytes); exit(EXIT_FAILURE); }
    if (actual_target_batch_bytes % 8) { fprintf(stderr, "[GASAL ERROR:] actual_target_batch_bytes=%d is not a multiple of 8\n", actual_target_batch_bytes); exit(EXIT_FAILURE); }
    // (storages created with isPacked keep their pages -- and host_max_*_batch_bytes -- in PACKED bytes, gasal_host_batch_fill_packed:
    //  a batch of `actual` unpacked-equivalent bytes occupies actual / 2 of them)
    const uint32_t host_q = params->isPacked2 ? actual_query_batch_bytes / 8 * 3 : params->isPacked ? actual_query_batch_bytes / 2 : actual_query_batch_bytes;
    const uint32_t host_t = params->isPacked2 ? actual_target_batch_bytes / 8 * 3 : params->isPacked ? actual_target_batch_bytes / 2 : actual_target_batch_bytes;
    if (host_q > s->host_max_query_batch_bytes) { fprintf(stderr, "[GASAL ERROR:] actual_query_batch_bytes(%d) > host_max_query_batch_bytes(%d)\n", actual_query_batch_bytes, s->host_max_query_batch_bytes); exit(EXIT_FAILURE); }
    if (host_t > s->host_max_target_batch_bytes) { fprintf(stderr, "[GASAL ERROR:] actual_target_batch_bytes(%d) > host_max_target_batch_bytes(%d)\n", actual_target_batch_bytes, s->host_max_target_batch_bytes); exit(EXIT_FAILURE); }
    if (actual_n_alns > s->host_max_n_alns) { fprintf(stderr, "[GASAL ERROR:] actual_n_alns(%d) > host_max_n_alns(%d)\n", actual_n_alns, s->host_max_n_alns); exit(EXIT_FAILURE); }

    // grow device buffers to a multiple of their current size (gasal_align.cu:71-133)
    auto grow_side = [&](uint32_t& cap, uint32_t need, uint8_t*& unpacked, uint32_t*& packed) {
        if (cap >= need) return;
        uint64_t nc = (uint64_t)cap * 2;
        while (nc < need) nc += cap;
        if (nc > 0xFFFFFFF0ull) nc = 0xFFFFFFF0ull & ~7ull;
        cap = (uint32_t)nc;
        CHK(agatha_amd_stream_synchronize(s->str));
        dev_free(unpacked);
        unpacked = dev_alloc<uint8_t>(cap + 16);
        if (params->isPacked) packed = (uint32_t*)unpacked;
        else { dev_free(packed); packed = dev_alloc<uint32_t>(cap / 8 + 4); }
    };
    grow_side(s->gpu_max_query_batch_bytes, actual_query_batch_bytes, s->unpacked_query_batch, s->packed_query_batch);
    grow_side(s->gpu_max_target_batch_bytes, actual_target_batch_bytes, s->unpacked_target_batch, s->packed_target_batch);
    if (s->gpu_max_n_alns < actual_n_alns) {
        uint32_t nc = s->gpu_max_n_alns * 2;
        while (nc < actual_n_alns) nc += s->gpu_max_n_alns;
        s->gpu_max_n_alns = nc;
        CHK(agatha_amd_stream_synchronize(s->str));
        free_device_meta(s);
        alloc_device_meta(s, nc);
        gasal_res_destroy_device(s->device_res, s->device_cpy);
        s->device_cpy = gasal_res_new_device_cpy(nc, params);
        s->device_res = gasal_res_new_device(s->device_cpy);
    }

    // H2D: every page of the two extensible host batches (gasal_align.cu:140-163)
    if (params->isPacked2) {
        // 2-bit + N-mask pages: the codes of all pages go to the front of the device's unpacked buffer (2 bytes per word), the mask
        // bytes behind them (at a quarter of its size); agatha_amd_unpack2 makes the 4-bit words the kernels read
        auto ship = [&](host_batch_t* head, uint8_t* stage, uint32_t cap_bytes, uint32_t actual_bytes, uint32_t* packed) {
            uint8_t* d_codes = stage; uint8_t* d_mask = stage + cap_bytes / 4;
            for (host_batch_t* p = head; p; p = p->next) {
                const uint32_t words = p->data_size / 3, w0 = p->offset / 3, cap_words = p->page_size / 3;
                if (!words) continue;
                CHK(agatha_amd_memcpy_h2d_async(s->str, d_codes + 2 * (size_t)w0, p->data, 2 * (size_t)words));
                CHK(agatha_amd_memcpy_h2d_async(s->str, d_mask + w0, p->data + 2 * (size_t)cap_words, words));
            }
            CHK(agatha_amd_unpack2(s->str, (const uint16_t*)d_codes, d_mask, actual_bytes, packed));
        };
        ship(s->extensible_host_unpacked_query_batch, s->unpacked_query_batch, s->gpu_max_query_batch_bytes & ~7u, actual_query_batch_bytes, s->packed_query_batch);
        ship(s->extensible_host_unpacked_target_batch, s->unpacked_target_batch, s->gpu_max_target_batch_bytes & ~7u, actual_target_batch_bytes, s->packed_target_batch);
    } else {
        for (host_batch_t* p = s->extensible_host_unpacked_query_batch; p; p = p->next)
            CHK(agatha_amd_memcpy_h2d_async(s->str, s->unpacked_query_batch + p->offset, p->data, p->data_size));
        for (host_batch_t* p = s->extensible_host_unpacked_target_batch; p; p = p->next)
            CHK(agatha_amd_memcpy_h2d_async(s->str, s->unpacked_target_batch + p->offset, p->data, p->data_size));
    }

    if (!params->isPacked && !params->isPacked2) {            // gasal_align.cu:174-185
        CHK(agatha_amd_pack(s->str, s->unpacked_query_batch, actual_query_batch_bytes, s->packed_query_batch));
        CHK(agatha_amd_pack(s->str, s->unpacked_target_batch, actual_target_batch_bytes, s->packed_target_batch));
    }
    const size_t mb = (size_t)actual_n_alns * sizeof(uint32_t);     // gasal_align.cu:191-194
    CHK(agatha_amd_memcpy_h2d_async(s->str, s->query_batch_lens, s->host_query_batch_lens, mb));
    CHK(agatha_amd_memcpy_h2d_async(s->str, s->target_batch_lens, s->host_target_batch_lens, mb));
    CHK(agatha_amd_memcpy_h2d_async(s->str, s->query_batch_offsets, s->host_query_batch_offsets, mb));
    CHK(agatha_amd_memcpy_h2d_async(s->str, s->target_batch_offsets, s->host_target_batch_offsets, mb));

    if (params->isReverseComplement) {          // gasal_align.cu:199-213
        if (params->isPacked || params->isPacked2) {
            fprintf(stderr, "[GASAL ERROR:] reverse/complement ops need the unpacked batch on the device (isPacked is set)\n");
            exit(EXIT_FAILURE);
        }
        CHK(agatha_amd_memcpy_h2d_async(s->str, s->query_op, s->host_query_op, actual_n_alns));
        CHK(agatha_amd_memcpy_h2d_async(s->str, s->target_op, s->host_target_op, actual_n_alns));
        CHK(agatha_amd_seq_ops(s->str, s->unpacked_query_batch, s->packed_query_batch, s->query_batch_lens,
                               s->query_batch_offsets, s->query_op, actual_n_alns));
        CHK(agatha_amd_seq_ops(s->str, s->unpacked_target_batch, s->packed_target_batch, s->target_batch_lens,
                               s->target_batch_offsets, s->target_op, actual_n_alns));
    }

    uint32_t max_q = 0, max_t = 0;      // length hints: let short batches use a narrower lane group
    for (uint32_t k = 0; k < actual_n_alns; k++) {
        max_q = std::max(max_q, s->host_query_batch_lens[k]);
        max_t = std::max(max_t, s->host_target_batch_lens[k]);
    }
    agatha_amd_scores sc = {g_scores.match, g_scores.mismatch, g_scores.gap_open, g_scores.gap_extend,
                            g_scores.slice_width, g_scores.z_threshold, g_scores.band_width};

    // sort + align; with -p the pair of events brackets them ON THIS STREAM (the reference brackets them on the
    // default stream and then device-synchronises, serialising its two streams: gasal_align.cu:219-236)
    if (params->traceback) {
        // alignment paths (extension; cigar / n_cigar_ops are declared and left NULL by the reference, gasal.h:91-92): the
        // result arrays and the code scratch are made on first use and grown with the batch
        const size_t cb = (size_t)s->gpu_max_query_batch_bytes + s->gpu_max_target_batch_bytes + 16;
        if (!s->host_res->cigar || !s->device_cpy->cigar || s->cigar_bytes < cb || s->cigar_alns < s->gpu_max_n_alns) {
            CHK(agatha_amd_stream_synchronize(s->str));
            pin_free(s->host_res->cigar); pin_free(s->host_res->n_cigar_ops);
            dev_free(s->device_cpy->cigar); dev_free(s->device_cpy->n_cigar_ops);
            s->host_res->cigar = pin_alloc<uint8_t>(cb);      s->host_res->n_cigar_ops = pin_alloc<uint32_t>(s->gpu_max_n_alns);
            s->device_cpy->cigar = dev_alloc<uint8_t>(cb);    s->device_cpy->n_cigar_ops = dev_alloc<uint32_t>(s->gpu_max_n_alns);
            s->cigar_bytes = cb; s->cigar_alns = s->gpu_max_n_alns;
        }
        const size_t per = agatha_amd_traceback_pair_bytes(max_q, max_t, &sc);
        if (per == 0) {
            fprintf(stderr, "[GASAL ERROR:] traceback: band width %d is too wide for sequences this long\n", g_scores.band_width);
            exit(EXIT_FAILURE);
        }
        // the whole batch in one pass if that takes at most 16 GiB, otherwise in passes of 16 GiB
        const uint32_t ppp = (uint32_t)std::max<size_t>(1, std::min<size_t>(actual_n_alns, ((size_t)16 << 30) / per));
        const size_t want = agatha_amd_traceback_scratch_bytes(actual_n_alns, max_q, max_t, &sc, ppp);
        if (s->tb_scratch_bytes < want) {
            CHK(agatha_amd_stream_synchronize(s->str));
            dev_free(s->tb_scratch);
            s->tb_scratch = dev_alloc<uint8_t>(want);
            s->tb_scratch_bytes = want;
        }
    }
    // (the gate: from here to the event behind the align call one thread at a time per device, so that ring[k] is recorded before batch k + LIMIT asks for it)
    const int gate_n = gate_limit();
    const int gate_dev = gate_n > 0 ? agatha_amd_get_device() : -1;
    DeviceGate* gate = (gate_dev >= 0 && gate_dev < kGateDevices) ? &g_gate[gate_dev] : nullptr;
    std::unique_lock<std::mutex> gate_lock;
    unsigned long long gate_k = 0;
    if (gate) {
        gate_lock = std::unique_lock<std::mutex>(gate->mu);
        gate_k = gate->ticket++;
        if (gate_k >= (unsigned long long)gate_n && gate->done[(gate_k - gate_n) % kGateRing]) CHK(agatha_amd_stream_wait_event(s->str, gate->done[(gate_k - gate_n) % kGateRing]));
    }
    if (params->print_out) CHK(agatha_amd_event_record(s->ev_begin, s->str));
    int rc = params->traceback
        ? agatha_amd_align_traceback(s->str, s->packed_query_batch, s->packed_target_batch, s->query_batch_lens,
                                     s->target_batch_lens, s->query_batch_offsets, s->target_batch_offsets, actual_n_alns,
                                     max_q, max_t, &sc, s->device_cpy->aln_score, s->device_cpy->query_batch_end,
                                     s->device_cpy->target_batch_end, s->device_cpy->cigar, s->device_cpy->n_cigar_ops,
                                     s->workspace, s->workspace_bytes, s->tb_scratch, s->tb_scratch_bytes)
        : agatha_amd_align(s->str, s->packed_query_batch, s->packed_target_batch, s->query_batch_lens,
                           s->target_batch_lens, s->query_batch_offsets, s->target_batch_offsets, actual_n_alns,
                           max_q, max_t, &sc, s->device_cpy->aln_score, s->device_cpy->query_batch_end,
                           s->device_cpy->target_batch_end, s->workspace, s->workspace_bytes);
    if (rc == AGATHA_AMD_EBAND) {
        fprintf(stderr, "[GASAL ERROR:] band width %d exceeds the largest supported band (%d) for sequences this long\n",
                g_scores.band_width, agatha_amd_max_band());
        exit(EXIT_FAILURE);
    }
    CHK(rc);
    if (gate) {
        void*& ev = gate->done[gate_k % kGateRing];
        if (!ev) CHK(agatha_amd_event_create(&ev));
        CHK(agatha_amd_event_record(ev, s->str));
        gate_lock.unlock();
    }
    if (params->print_out) { CHK(agatha_amd_event_record(s->ev_end, s->str)); s->timing_pending = 1; s->timing_params = params; s->timing_n_alns = actual_n_alns; }
    // the int16 kernel's guard counters of this batch, on their way to pinned memory behind the kernels (no wait; looked at in
    // gasal_is_aln_async_done): pairs it ended or refused to resume because their saved state was not what it should be -- the int32 kernel
    // has redone them, the results are right, and the user is told
    if (s->guard_host) CHK(agatha_amd_guard_stats(s->str, s->workspace, actual_n_alns, s->guard_host, 0));

    if (params->start_pos) {
        // start positions (extension; the reference copies them back only if they are not NULL, gasal_align.cu:256-260, and they
        // always are): the same extension run backwards from every end cell, agatha_amd_align_starts
        const size_t need = agatha_amd_starts_scratch_bytes(s->gpu_max_query_batch_bytes, s->gpu_max_target_batch_bytes, s->gpu_max_n_alns);
        if (s->starts_scratch_bytes < need) {
            CHK(agatha_amd_stream_synchronize(s->str));
            dev_free(s->starts_scratch);
            s->starts_scratch = dev_alloc<uint8_t>(need);
            s->starts_scratch_bytes = need;
        }
        if (!s->device_cpy->query_batch_start || !s->host_res->query_batch_start) {
            fprintf(stderr, "[GASAL ERROR:] start_pos was set after the result objects were created\n");
            exit(EXIT_FAILURE);
        }
        CHK(agatha_amd_align_starts(s->str, s->packed_query_batch, s->packed_target_batch, s->query_batch_offsets,
                                    s->target_batch_offsets, actual_n_alns, actual_query_batch_bytes, actual_target_batch_bytes,
                                    max_q, max_t, &sc, s->device_cpy->query_batch_end, s->device_cpy->target_batch_end,
                                    s->device_cpy->query_batch_start, s->device_cpy->target_batch_start, s->workspace,
                                    s->workspace_bytes, s->starts_scratch, s->starts_scratch_bytes));
    }

    const size_t rb = (size_t)actual_n_alns * sizeof(int32_t);      // gasal_align.cu:253-266
    CHK(agatha_amd_memcpy_d2h_async(s->str, s->host_res->aln_score, s->device_cpy->aln_score, rb));
    CHK(agatha_amd_memcpy_d2h_async(s->str, s->host_res->query_batch_end, s->device_cpy->query_batch_end, rb));
    CHK(agatha_amd_memcpy_d2h_async(s->str, s->host_res->target_batch_end, s->device_cpy->target_batch_end, rb));
    if (s->host_res->query_batch_start && s->device_cpy->query_batch_start)
        CHK(agatha_amd_memcpy_d2h_async(s->str, s->host_res->query_batch_start, s->device_cpy->query_batch_start, rb));
    if (s->host_res->target_batch_start && s->device_cpy->target_batch_start)
        CHK(agatha_amd_memcpy_d2h_async(s->str, s->host_res->target_batch_start, s->device_cpy->target_batch_start, rb));
    if (params->traceback) {
        CHK(agatha_amd_memcpy_d2h_async(s->str, s->host_res->n_cigar_ops, s->device_cpy->n_cigar_ops, rb));
        CHK(agatha_amd_memcpy_d2h_async(s->str, s->host_res->cigar, s->device_cpy->cigar,
                                        (size_t)actual_query_batch_bytes + actual_target_batch_bytes));
    }
    s->is_free = 0;
}

// -2: nothing launched, -1: still running, 0: done (results valid, stream free again)   gasal_align.cu:276-292
int gasal_is_aln_async_done(gasal_gpu_storage_t* s)
{
    if (s->is_free == 1) return -2;
    const int q = agatha_amd_stream_query(s->str);
    if (q == 1) return -1;
    if (q < 0) die_hip(q, __LINE__);
    // -p: the library itself appends the batch's kernel milliseconds to params->raw_file, as the reference does inside
    // gasal_aln_async (gasal_align.cu:218-236).  The reference blocks there (cudaDeviceSynchronize); here the call stays
    // asynchronous and the line is written when the batch is seen to be complete.
    if (s->timing_pending) {
        float ms = 0.f;
        CHK(agatha_amd_event_elapsed_ms(s->ev_begin, s->ev_end, &ms));
        Parameters* p = (Parameters*)s->timing_params;
        if (p) { std::lock_guard<std::mutex> lock(g_raw_mutex); p->raw_file << ms << std::endl; }
        // (measurement aid, not part of the reference's protocol: with AGATHA_AMD_RAW_STATS=<file> every batch also leaves a
        //  line "n_alns kernel_ms value_steps key_steps started_over went_back_to_checkpoint taken_over" there --
        //  agatha_amd_step_stats; taken_over counts the pairs a lane group took from a neighbour that did not show up in time)
        static const char* stats_path = getenv("AGATHA_AMD_RAW_STATS");
        // (not with start positions: agatha_amd_align_starts has by now run the batch again, backwards, on the same workspace, and the
        //  counters would be that pass's, not the timed kernel's)
        if (stats_path && *stats_path && !(p && p->start_pos)) {
            unsigned int st[40] = {0};
            if (agatha_amd_step_stats(s->str, s->workspace, s->timing_n_alns, st) == 0) {
                std::lock_guard<std::mutex> lock(g_raw_mutex);
                if (FILE* f = fopen(stats_path, "a")) {
                    fprintf(f, "%u %.4f %u %u %u %u %u\n", s->timing_n_alns, ms, st[0], st[1], st[2], st[15], st[14]);
                    fclose(f);
                }
            }
        }
        s->timing_pending = 0;
        s->timing_params = nullptr;
    }
    if (s->guard_host && (s->guard_host[0] || s->guard_host[1])) {
        fprintf(stderr, "[GASAL WARNING:] the packed-int16 kernel ended %u pair(s) whose step counter had run past their last step and refused %u saved "
                        "state(s) that failed their check (%u poisoned on purpose: debug option poison_state); the int32 kernel redid those pairs and the "
                        "results are right, but device memory of this library was overwritten or its state machine has a bug -- please report\n",
                s->guard_host[0], s->guard_host[1], s->guard_host[2]);
        memset(s->guard_host, 0, 4 * sizeof(unsigned int));
    }
    gasal_host_batch_reset(s);
    s->is_free = 1;
    s->current_n_alns = 0;
    return 0;
}
