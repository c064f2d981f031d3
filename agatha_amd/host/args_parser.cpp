// args_parser.cpp -- the `Parameters` class of the GASAL/AGAThA CLI contract (include/gasal_header.h).
// Same flags, defaults and positional rules as the reference (AGAThA/src/args_parser.cpp:8-229):
//   manual [-m a] [-x b] [-q go] [-r ge] [-s sw] [-z z] [-w w] [-b blocks] [-t threads] [-a alns/batch]
//          [-n host_threads] [-p] <file1.fasta> <file2.fasta> [rawlog if -p]
// Options are scanned over argv[1 .. argc-4]; the last two (three with -p) arguments are positional; argc < 4 is
// refused; unknown single letters are ignored.  -b/-t are accepted and ignored by the HIP engine (its launch shape
// is derived from the band); -c (extension) enables the reverse/complement op codes of the FASTA headers; -g N
// (extension) spreads the host threads over N GPUs; -k (extension) packs on the host (isPacked storages, ctors.cpp:65-73),
// -K (extension) into the 2-bit + N-mask format;
// -S (extension) fills the start members of the results (the reference declares them and leaves them NULL).
#include "../../include/gasal_header.h"

Parameters::Parameters(int argc_, char** argv_)
    : sa(2), sb(4), gapo(4), gape(2), print_out(0), n_threads(1), slice_width(3), z_threshold(400), band_width(751),
      kernel_block_num(256), kernel_thread_num(256), kernel_align_num(8192), isPacked(false),
      isReverseComplement(false), start_pos(0), traceback(0), n_gpus(1), isPacked2(false), argc(argc_), argv(argv_)
{
}

Parameters::~Parameters()
{
    query_batch_fasta.close();
    target_batch_fasta.close();
    raw_file.close();
}

void Parameters::print()
{
    std::cerr << "sa=" << sa << " , sb=" << sb << " , gapo=" << gapo << " , gape=" << gape << std::endl;
    std::cerr << "slice_width=" << slice_width << ", z_threshold=" << z_threshold << ", band_width=" << band_width << std::endl;
    std::cerr << "kernel launch: block_num=" << kernel_block_num << ", thread_num=" << kernel_thread_num
              << ", align_num=" << kernel_align_num << std::endl;
    std::cerr << "print_out=" << print_out << " , n_threads=" << n_threads << std::endl;
    std::cerr << std::boolalpha << "isPacked = " << isPacked << ", isPacked2 = " << isPacked2 << std::endl;
    std::cerr << "query_batch_fasta_filename=" << query_batch_fasta_filename
              << " , target_batch_fasta_filename=" << target_batch_fasta_filename << std::endl;
}

void Parameters::failure(fail_type f)
{
    switch (f) {
        case NOT_ENOUGH_ARGS:
            std::cerr << "Not enough Parameters. Required: [options] file1.fasta file2.fasta. See help (--help, -h) for usage. " << std::endl;
            break;
        case WRONG_ARG:
            std::cerr << "Wrong argument. See help (--help, -h) for usage. " << std::endl;
            break;
        case WRONG_FILES:
            std::cerr << "File error: either a file doesn't exist, or cannot be opened." << std::endl;
            break;
        default:
            break;
    }
    exit(1);
}

void Parameters::help()
{
    std::cerr << "Usage: ./manual [-m] [-x] [-q] [-r] [-s] [-z] [-w] [-b] [-t] [-a] [-p] [-n] <query_batch.fasta> <target_batch.fasta> [raw.log]" << std::endl;
    std::cerr << "Options: -m INT    match score [" << sa << "]" << std::endl;
    std::cerr << "         -x INT    mismatch penalty [" << sb << "]" << std::endl;
    std::cerr << "         -q INT    gap open penalty [" << gapo << "]" << std::endl;
    std::cerr << "         -r INT    gap extension penalty [" << gape << "]" << std::endl;
    std::cerr << "         -s INT    slice width: z-drop is tested every s block anti-diagonals [" << slice_width << "]" << std::endl;
    std::cerr << "         -z INT    z-drop threshold, < 0 disables [" << z_threshold << "]" << std::endl;
    std::cerr << "         -w INT    band width [" << band_width << "]" << std::endl;
    std::cerr << "         -b, -t    accepted for compatibility (CUDA launch shape); ignored" << std::endl;
    std::cerr << "         -a INT    alignments per batch [" << kernel_align_num << "]" << std::endl;
    std::cerr << "         -p        print the alignment results; append kernel ms per batch to raw.log" << std::endl;
    std::cerr << "         -n INT    number of CPU threads [" << n_threads << "]" << std::endl;
    std::cerr << "         -c        apply the reverse/complement codes of the FASTA header characters (> < / +)" << std::endl;
    std::cerr << "         -k        pack the sequences on the host and ship pre-packed batches (isPacked): half the H2D bytes" << std::endl;
    std::cerr << "         -K        the same in the 2-bit + N-mask format (isPacked2; A, C, G, T, N only): 3/8 of the H2D bytes" << std::endl;
    std::cerr << "         -S        also compute and print the start positions (query_batch_start / target_batch_start):" << std::endl;
    std::cerr << "                   where the best alignment ENDING in the end cell begins (local-style trimming; not the first cell of the -T path," << std::endl;
    std::cerr << "                   which is the extension alignment and always starts at the origin)" << std::endl;
    std::cerr << "         -T        also compute and print the alignment paths (cigar / n_cigar_ops)" << std::endl;
    std::cerr << "         -g INT    spread the CPU threads over this many GPUs [" << n_gpus << "]" << std::endl;
    std::cerr << "         --help, -h : displays this message." << std::endl;
    std::cerr << "Single-pack multi-Parameters (e.g. -sp) is not supported." << std::endl;
}

void Parameters::parse()
{
    for (int c = 1; c < argc; c++) {
        const std::string cur(argv[c]);
        if (cur == "--help" || cur == "-h") { help(); exit(0); }
    }
    if (argc < 4) failure(NOT_ENOUGH_ARGS);

    int c;
    auto next_int = [&](int& pos) { pos++; return std::stoi(std::string(argv[pos])); };
    for (c = 1; c < argc - 3; c++) {
        const std::string cur(argv[c]);
        if (cur.size() >= 2 && cur[0] == '-' && cur[1] == '-') continue;
        if (cur.empty() || cur[0] != '-') failure(WRONG_ARG);
        if (cur.length() != 2) failure(WRONG_ARG);
        switch (cur[1]) {
            case 'm': sa = next_int(c); break;
            case 'x': sb = next_int(c); break;
            case 'q': gapo = next_int(c); break;
            case 'r': gape = next_int(c); break;
            case 'p': print_out = 1; break;
            case 'c': isReverseComplement = true; break;
            case 'n': n_threads = next_int(c); break;
            case 'g': n_gpus = next_int(c); break;
            case 'k': isPacked = true; break;
            case 'K': isPacked2 = true; break;
            case 'S': start_pos = 1; break;
            case 'T': traceback = 1; break;
            case 's': slice_width = next_int(c); break;
            case 'z': z_threshold = next_int(c); break;
            case 'w': band_width = next_int(c); break;
            case 'b': kernel_block_num = next_int(c); break;
            case 't': kernel_thread_num = next_int(c); break;
            case 'a': kernel_align_num = next_int(c); break;
            default: break;          // unknown letters are silently ignored, as in the reference
        }
    }
    query_batch_fasta_filename = std::string(argv[c]);
    c++;
    target_batch_fasta_filename = std::string(argv[c]);
    if (print_out) {
        c++;
        if (c < argc) raw_filename = std::string(argv[c]);
    }
    fileopen();
}

void Parameters::fileopen()
{
    query_batch_fasta.open(query_batch_fasta_filename, std::ifstream::in);
    if (!query_batch_fasta) failure(WRONG_FILES);
    target_batch_fasta.open(target_batch_fasta_filename);
    if (!target_batch_fasta) failure(WRONG_FILES);
    if (print_out && !raw_filename.empty()) raw_file.open(raw_filename, std::ios::app);
}
