// test_prog.cpp -> `manual`: the AGAThA command line on the MI355X engine.
//
// Caller contract of the reference's AGAThA/test_prog/test_prog.cpp:21-410: two FASTA files with the same
// number of records (record k of file 1 = DP rows / "query_batch", record k of file 2 = DP columns /
// "target_batch"), header character > < / + = op code 0..3, sequence lines concatenated; pairs are cut into
// batches of -a alignments, each batch goes through gasal_host_batch_fill + gasal_aln_async on one of
// NB_STREAMS storages per host thread; with -p every pair prints
//     <score>\tquery_batch_end=<q>\ttarget_batch_end=<t>
// in input order within a batch; the library appends one kernel-milliseconds line per batch to the raw log
// (gasal_is_aln_async_done, as the reference's gasal_aln_async does: gasal_align.cu:218-236).  -g N (extension) spreads
// the host threads over N GPUs with gasal_set_device (interfaces.cpp:86-116, the call the reference leaves commented
// out at test_prog.cpp:31).
// Own reader (each file is parsed on its own, so records with different line counts stay in step).
// Measurement aids (environment, not flags -- the reference's command line stays as it is): AGATHA_AMD_REPEAT=R runs the pairs of
// the two files R times over, as if the files were R times as long (a sustained feed without gigabytes of FASTA);
// AGATHA_AMD_LOOP_STATS=<file> appends "seconds pairs batches threads" of the batch loop (first fill to last result: host fill,
// H2D, pack, sort, align, D2H, stream-overlapped -- SURVEY.md 8(d)'s wall time; the FASTA parse before it is not in it).
#include "../../include/gasal_header.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>

#include <chrono>
#include <thread>

#include <vector>

#define NB_STREAMS 2

struct FastaSet {
    std::vector<std::string> seqs;
    std::vector<uint8_t> ops;
};

static void read_fasta(std::ifstream& in, FastaSet& out)
{
    static const char line_starts[] = "></+";      // op = index: forward, reverse, complement, reverse-complement
    std::string line;
    bool open_record = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        const char* hit = strchr(line_starts, line[0]);
        if (hit && *hit) {
            out.ops.push_back((uint8_t)(hit - line_starts));
            out.seqs.emplace_back();
            open_record = true;
        } else if (open_record) {
            out.seqs.back() += line;
        } else {
            std::cerr << "Batch1 and target_batch files should be fasta having same number of sequences" << std::endl;
            exit(EXIT_FAILURE);
        }
    }
}

int main(int argc, char** argv)
{
    Parameters* args = new Parameters(argc, argv);
    args->parse();
    const int print_out = args->print_out;
    const int n_threads = args->n_threads > 0 ? args->n_threads : 1;

    gasal_subst_scores sub_scores;
    sub_scores.match = args->sa;
    sub_scores.mismatch = args->sb;
    sub_scores.gap_open = args->gapo;
    sub_scores.gap_extend = args->gape;
    sub_scores.slice_width = args->slice_width;
    sub_scores.z_threshold = args->z_threshold;
    sub_scores.band_width = args->band_width;
    gasal_copy_subst_scores(&sub_scores);

    FastaSet Qs, Ts;
    read_fasta(args->query_batch_fasta, Qs);
    read_fasta(args->target_batch_fasta, Ts);
    if (Qs.seqs.size() != Ts.seqs.size() || Qs.seqs.empty()) {
        std::cerr << "Batch1 and target_batch files should be fasta having same number of sequences" << std::endl;
        exit(EXIT_FAILURE);
    }
    const int file_seqs = (int)Qs.seqs.size();
    int repeat = 1;
    if (const char* r = getenv("AGATHA_AMD_REPEAT")) repeat = std::max(1, atoi(r));
    const int total_seqs = file_seqs * repeat;
    uint32_t max_q = 0, max_t = 0;
    for (int i = 0; i < file_seqs; i++) {
        max_q = std::max<uint32_t>(max_q, (uint32_t)Qs.seqs[i].size());
        max_t = std::max<uint32_t>(max_t, (uint32_t)Ts.seqs[i].size());
    }
    const uint32_t maximum_sequence_length = std::max(max_q, max_t);

    // equal split of the pairs over the host threads (test_prog.cpp:195-203)
    std::vector<int> thread_first(n_threads), thread_count(n_threads);
    const int per_thread = (int)ceil((double)total_seqs / n_threads);
    for (int t = 0, done = 0; t < n_threads; t++) {
        thread_first[t] = done;
        thread_count[t] = std::max(0, std::min(per_thread, total_seqs - done));
        done += thread_count[t];
    }

    omp_set_num_threads(n_threads);
    std::vector<gasal_gpu_storage_v> vecs(n_threads);
    const int n_gpus = args->n_gpus > 0 ? args->n_gpus : 1;
    for (int t = 0; t < n_threads; t++) {
        if (n_gpus > 1) gasal_set_device(t % n_gpus, false);      // this thread's buffers and streams live on its GPU
        vecs[t] = gasal_init_gpu_storage_v(NB_STREAMS);
        gasal_init_streams(&vecs[t], (int)max_q + 7, (int)max_t + 7, (int32_t)maximum_sequence_length, args);
    }

    const auto loop_t0 = std::chrono::steady_clock::now();
    long total_batches = 0;
#pragma omp parallel reduction(+ : total_batches)
    {
        const int tid = omp_get_thread_num();
        if (n_gpus > 1) gasal_set_device(tid % n_gpus, false);
        const int n_seqs = thread_count[tid];
        const int n_batches = (int)ceil((double)n_seqs / args->kernel_align_num);
        int next = thread_first[tid], seqs_done = 0, batches_done = 0;
        struct Slot { gasal_gpu_storage_t* st; int n; } slot[NB_STREAMS];
        for (int z = 0; z < NB_STREAMS; z++) { slot[z].st = &vecs[tid].a[z]; slot[z].n = 0; }

        while (batches_done < n_batches) {
            bool progress = false;      // (round 6) an iteration that neither launched nor finished a batch backs off: four threads spinning on the stream queries
                                        // of eight storages fought the launching threads for the runtime's locks -- "-n 4" ran at the rate of "-n 1"
            int z = 0;
            while (z < NB_STREAMS && slot[z].st->is_free != 1) z++;
            if (seqs_done < n_seqs && z < NB_STREAMS) {
                progress = true;
                gasal_gpu_storage_t* st = slot[z].st;
                uint32_t qidx = 0, tidx = 0;
                int j = 0;
                const int first = next;
                for (; seqs_done < n_seqs && j < args->kernel_align_num; j++, seqs_done++, next++) {
                    const int rec = next % file_seqs;          // (AGATHA_AMD_REPEAT: the files' pairs over again)
                    st->current_n_alns++;
                    if (st->current_n_alns > st->host_max_n_alns) gasal_host_alns_resize(st, st->host_max_n_alns * 2, args);
                    st->host_query_batch_offsets[j] = qidx;
                    st->host_target_batch_offsets[j] = tidx;
                    if (args->isPacked2) {      // -K: 2-bit codes + N mask packed on the host, expanded on the device
                        qidx = gasal_host_batch_fill_packed2(st, qidx, Qs.seqs[rec].c_str(), (uint32_t)Qs.seqs[rec].size(), QUERY);
                        tidx = gasal_host_batch_fill_packed2(st, tidx, Ts.seqs[rec].c_str(), (uint32_t)Ts.seqs[rec].size(), TARGET);
                    } else if (args->isPacked) {       // -k: 4-bit words packed on the host, no pack kernel (ctors.cpp:65-73)
                        qidx = gasal_host_batch_fill_packed(st, qidx, Qs.seqs[rec].c_str(), (uint32_t)Qs.seqs[rec].size(), QUERY);
                        tidx = gasal_host_batch_fill_packed(st, tidx, Ts.seqs[rec].c_str(), (uint32_t)Ts.seqs[rec].size(), TARGET);
                    } else {
                        qidx = gasal_host_batch_fill(st, qidx, Qs.seqs[rec].c_str(), (uint32_t)Qs.seqs[rec].size(), QUERY);
                        tidx = gasal_host_batch_fill(st, tidx, Ts.seqs[rec].c_str(), (uint32_t)Ts.seqs[rec].size(), TARGET);
                    }
                    st->host_query_batch_lens[j] = (uint32_t)Qs.seqs[rec].size();
                    st->host_target_batch_lens[j] = (uint32_t)Ts.seqs[rec].size();
                }
                if (repeat == 1) {
                    gasal_op_fill(st, Qs.ops.data() + first, (uint32_t)j, QUERY);
                    gasal_op_fill(st, Ts.ops.data() + first, (uint32_t)j, TARGET);
                } else {                                        // (a batch may wrap around the end of the files)
                    std::vector<uint8_t> qo_(j), to_(j);
                    for (int k = 0; k < j; k++) { qo_[k] = Qs.ops[(first + k) % file_seqs]; to_[k] = Ts.ops[(first + k) % file_seqs]; }
                    gasal_op_fill(st, qo_.data(), (uint32_t)j, QUERY);
                    gasal_op_fill(st, to_.data(), (uint32_t)j, TARGET);
                }
                slot[z].n = j;
                gasal_aln_async(st, qidx, tidx, (uint32_t)j, args);
                st->current_n_alns = 0;
            }
            for (z = 0; z < NB_STREAMS; z++) {
                if (gasal_is_aln_async_done(slot[z].st) == 0) {
                    if (print_out) {
#pragma omp critical
                        {
                            const gasal_res_t* r = slot[z].st->host_res;
                            for (int j = 0; j < slot[z].n; j++) {
                                std::cout << r->aln_score[j] << "\tquery_batch_end=" << r->query_batch_end[j]
                                          << "\ttarget_batch_end=" << r->target_batch_end[j];
                                if (args->start_pos)        // -S (extension)
                                    std::cout << "\tquery_batch_start=" << r->query_batch_start[j] << "\ttarget_batch_start="
                                              << r->target_batch_start[j];
                                if (args->traceback) {      // -T (extension): the path as CIGAR text (=, X, D, I)
                                    const gasal_gpu_storage_t* st = slot[z].st;
                                    const uint32_t nops = r->n_cigar_ops[j];
                                    std::cout << "\tcigar=";
                                    if (nops == 0xFFFFFFFFu) std::cout << "!";
                                    else if (nops == 0) std::cout << "*";
                                    else {
                                        const uint8_t* c = r->cigar + st->host_query_batch_offsets[j] + st->host_target_batch_offsets[j];
                                        uint32_t run = 0, op = 4;
                                        for (uint32_t b = 0; b <= nops; b++) {
                                            const uint32_t o = b < nops ? (c[b] & 3u) : 5u;
                                            if (o != op) { if (run) std::cout << run << "=XDI"[op]; run = 0; op = o; }
                                            if (b < nops) run += c[b] >> 2;
                                        }
                                    }
                                }
                                std::cout << std::endl;
                            }
                        }
                    }
                    batches_done++;
                    progress = true;
                }
            }
            if (!progress) std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        total_batches += batches_done;
    }
    if (const char* path = getenv("AGATHA_AMD_LOOP_STATS")) {
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - loop_t0).count();
        if (FILE* f = fopen(path, "a")) { fprintf(f, "%.6f %d %ld %d\n", sec, total_seqs, total_batches, n_threads); fclose(f); }
    }

    for (int t = 0; t < n_threads; t++) {
        if (n_gpus > 1) gasal_set_device(t % n_gpus, false);
        gasal_destroy_streams(&vecs[t], args);
        gasal_destroy_gpu_storage_v(&vecs[t]);
    }
    delete args;
    return 0;
}
