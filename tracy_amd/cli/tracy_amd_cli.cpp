// tracy_amd_cli.cpp -- `tracy align` and `tracy decompose` rebuilt on the C ABI (SURVEY.md section 8(f)
// rank 1): the option sets, progress lines, exit codes and output files of /root/reference/src/sage.h:58-356
// and indigo.h:42-455, plus a --batch manifest mode that pushes many traces through ONE device batch
// (tracyhip_align_traces / tracyhip_decompose_traces).
//
//   tracy_amd_cli align     [options] -r reference.fa|wildtype.ab1 trace.ab1
//   tracy_amd_cli decompose [options] -r reference.fa trace.ab1
//   tracy_amd_cli <cmd>     [options] --batch manifest.tsv     lines: trace <TAB> reference <TAB> outprefix
//   tracy_amd_cli assemble  [options] [-r reference.fa] trace1.ab1 trace2.ab1 ...   (assemble_cli.inc)
//
// Host stages (as in the reference): ABIF/SCF parsing, basecalling, trimming estimate, profiles, file
// writers.  Device stages: every Gotoh DP, orientation, trimReferenceSlice, alignment rows.  There is no
// CPU fallback: without a GPU the command fails with the library's error text.
// Both commands also take an indexed genome (gzip-compressed multi-FASTA): the trace is anchored by k-mer votes
// (seed.hpp, fmindex.h:173-326) and aligned against the window around the hit.
// Not built: --annotate (needs the network), BCF output (no htslib).  Variants (-v) are written as
// VCF text because htslib (BCF) is not available.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <sched.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <sys/resource.h>
#include <thread>
#include <vector>

#include "../../include/tracy_hip.h"
#include "../host/assemble_out.hpp"
#include "../host/consensus_out.hpp"
#include "../host/bcf_out.hpp"
#include "../host/indigo_out.hpp"
#include "../host/sage_out.hpp"
#include "../host/seed.hpp"

using namespace tracy_amd;

namespace {

struct SageConfig {  // sage.h:37-56 + the extra fields of IndigoConfig (indigo.h:16-40)
  uint16_t linelimit = 60, trimLeft = 50, trimRight = 50, maxindel = 1000, madc = 5, qualCut = 45, kmer = 15, minKmerSupport = 3;
  float pratio = 0.33f, trimStringency = 0;
  int32_t gapopen = -10, gapext = -4, match = 3, mismatch = -5;
  bool callvariants = false;
  std::string outprefix = "out", genome, ab, batch, annotate;
  int device = 0;
  std::string devices = "0";  // -d: one ordinal, a comma-separated list, or "all" (batches are cut into blocks, one per GPU)
  uint32_t threads = 0;        // --threads: host threads of the --batch stages (read, basecall, profile, writers); 0 = every core this process may use
};

struct Job {
  std::string trace_path, ref_path, outprefix;
  Trace tr;
  BaseCalls bc;
  uint32_t trimLeft = 0, trimRight = 0;
  Profile full;
  ReferenceSlice rs;
  std::string fasta;         // filetype 1: the loaded record (forward strand); filetype 0: the oriented window
  uint32_t slice_start = 0;  // filetype 0: offset of the window in its contig
  Profile wt_fwd;            // filetype 2: wildtype profile
  std::string wt_primary;
  int32_t score = 0;
  AlignRows rows;
  bool ok = false;
  // decompose
  std::string primary, secondary, secdecomp;  // decomposed basecalls
  int32_t status = 0;
  tracyhip_decomp_status dstatus{};
  AlleleReport rep;
};

// cores this process may really use: the affinity mask, capped by the cgroup CPU quota of the container
uint32_t usable_cores() {
  uint32_t n = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = (uint32_t)CPU_COUNT(&set);
  std::ifstream q("/sys/fs/cgroup/cpu.max");
  std::string quota;
  long period = 0;
  if (q >> quota >> period && quota != "max" && period > 0) {
    const long cores = std::atol(quota.c_str()) / period;
    if (cores >= 1 && (uint32_t)cores < n) n = (uint32_t)cores;
  }
  return n ? n : 1;
}
// fn(i) for i in [0, n) on `threads` host threads (work handed out one index at a time: traces differ in length)
template <class Fn>
void for_each_index(uint32_t n, uint32_t threads, Fn fn) {
  if (threads <= 1 || n <= 1) { for (uint32_t i = 0; i < n; ++i) fn(i); return; }
  std::atomic<uint32_t> next(0);
  std::vector<std::thread> th;
  const uint32_t nth = threads < n ? threads : n;
  for (uint32_t t = 0; t < nth; ++t)
    th.emplace_back([&]() { for (uint32_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); });
  for (auto& x : th) x.join();
}
const std::chrono::steady_clock::time_point g_process_start = std::chrono::steady_clock::now();
// TRACY_AMD_CLI_TIMERS: seconds of host-thread time inside the stages, summed over the threads (cpu_*: where a stage's core time goes)
struct CpuPhases {
  enum { LOAD, BASECALL, ABIF_TXT, PROFILE, REFERENCE, PACK, DEVICE_CALL, UNPACK, VARIANTS, PAD, SMALL_FILES, JSON, COUNT };
  std::atomic<uint64_t> ns[COUNT];
  bool on = getenv("TRACY_AMD_CLI_TIMERS") != nullptr;
  CpuPhases() { for (auto& x : ns) x = 0; }
  static const char* name(int i) {
    static const char* n[COUNT] = {"load", "basecall", "abif_txt", "profile", "reference", "pack", "device_call", "unpack", "variants", "pad", "small_files", "json"};
    return n[i];
  }
};
inline CpuPhases& cpu_phases() { static CpuPhases p; return p; }
struct PhaseClock {  // lap(i): the time since the last lap goes to phase i
  std::chrono::steady_clock::time_point t;
  const bool on;
  PhaseClock() : on(cpu_phases().on) { if (on) t = std::chrono::steady_clock::now(); }
  void lap(int phase) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    cpu_phases().ns[phase] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t).count();
    t = t1;
  }
};
// TRACY_AMD_CLI_TIMERS=1: one line on stderr with the wall time of the host and device stages of a --batch run
struct StageTimes {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), start = t0;
  std::vector<std::pair<std::string, double>> v;
  std::mutex m;
  void mark(const char* what) {
    const auto t1 = std::chrono::steady_clock::now();
    v.emplace_back(what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  }
  // stages that run side by side (run_blocks): the seconds each was busy, summed over its blocks
  void add(const char* what, double seconds) {
    std::lock_guard<std::mutex> lk(m);
    for (auto& e : v)
      if (e.first == what) { e.second += seconds; return; }
    v.emplace_back(what, seconds);
  }
  // peak resident set of THIS program: VmHWM of /proc/self/status belongs to the address space, which exec() replaces -- ru_maxrss is
  // inherited across fork + exec and reports the parent's high-water mark when that was larger (a Python process that loaded torch)
  static double peak_rss_mb() {
    double mb = 0;
    if (FILE* f = std::fopen("/proc/self/status", "r")) {
      char line[256];
      while (std::fgets(line, sizeof line, f))
        if (!std::strncmp(line, "VmHWM:", 6)) { mb = std::strtod(line + 6, nullptr) / 1024.0; break; }
      std::fclose(f);
    }
    if (mb == 0) {  // no procfs: the inherited figure is better than none
      struct rusage ru;
      getrusage(RUSAGE_SELF, &ru);
      mb = ru.ru_maxrss / 1024.0;
    }
    return mb;
  }
  void report(uint32_t traces, uint32_t threads) {
    if (!getenv("TRACY_AMD_CLI_TIMERS")) return;
    std::cerr << "timers: traces " << traces << " host_threads " << threads;
    for (auto const& e : v) std::cerr << " " << e.first << " " << e.second;
    for (int i = 0; i < CpuPhases::COUNT; ++i)
      if (cpu_phases().ns[i]) std::cerr << " cpu_" << CpuPhases::name(i) << "_s " << 1e-9 * (double)cpu_phases().ns[i];
    std::cerr << " before_stages_s " << std::chrono::duration<double>(start - g_process_start).count();
    std::cerr << " wall_s " << std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() << " peak_rss_mb " << peak_rss_mb() << std::endl;
  }
};
// the end of a --batch command: every file is written and closed; what is left is handing back a few GB of host and device memory
// and unloading the HIP runtime piece by piece (0.1-0.2 s) -- the operating system does that at once for a process that just ends
[[noreturn]] inline void end_process(int rc) {
  std::cout.flush();
  std::cerr.flush();
  std::fflush(nullptr);
  _exit(rc);
}
struct Stopwatch {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// --batch: the manifest in blocks through three stages in flight.  Host threads read / basecall / profile block k + 1 (`prep`) and
// write the files of block k - 1 (`write`) while the device works on block k (`device`, on the calling thread, in manifest order);
// a block's traces are released once its files are written, so memory is bounded by the blocks in flight (four), not by the
// manifest.  device() returns false to abort the command.  One block = the reference's one-trace-per-process flow, stage after stage.
constexpr uint32_t kDefaultBlock = 2000;
inline uint32_t block_size() {
  const char* e = getenv("TRACY_AMD_CLI_BLOCK");
  const long v = e ? std::atol(e) : 0;
  return v >= 1 ? (uint32_t)v : kDefaultBlock;
}
template <class Prep, class Device_, class Write>
bool run_blocks(uint32_t njobs, uint32_t block, Prep prep, Device_ device, Write write) {
  const uint32_t nb = (njobs + block - 1) / block;
  if (nb <= 1) {
    prep(0u, njobs);
    if (!device(0u, njobs)) return false;
    write(0u, njobs);
    return true;
  }
  std::mutex m;
  std::condition_variable cv;
  uint32_t prepared = 0, deviced = 0, written = 0;  // blocks done by each stage
  bool abort_ = false;
  auto range = [&](uint32_t b, uint32_t& lo, uint32_t& hi) { lo = b * block; hi = std::min(njobs, lo + block); };
  std::thread tp([&]() {
    for (uint32_t b = 0; b < nb; ++b) {
      {  // at most two prepared blocks wait for the device, and two for their writers
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return abort_ || b < written + 4u; });
        if (abort_) return;
      }
      uint32_t lo, hi;
      range(b, lo, hi);
      prep(lo, hi);
      { std::lock_guard<std::mutex> lk(m); prepared = b + 1; }
      cv.notify_all();
    }
  });
  std::thread tw([&]() {
    for (uint32_t b = 0; b < nb; ++b) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return abort_ || deviced > b; });
        if (abort_) return;
      }
      uint32_t lo, hi;
      range(b, lo, hi);
      write(lo, hi);
      { std::lock_guard<std::mutex> lk(m); written = b + 1; }
      cv.notify_all();
    }
  });
  bool ok = true;
  for (uint32_t b = 0; b < nb && ok; ++b) {
    {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return prepared > b; });
    }
    uint32_t lo, hi;
    range(b, lo, hi);
    ok = device(lo, hi);
    { std::lock_guard<std::mutex> lk(m); if (ok) deviced = b + 1; else abort_ = true; }
    cv.notify_all();
  }
  tp.join();
  tw.join();
  return ok;
}

std::string stamp() {  // boost::posix_time::to_simple_string(second_clock::local_time())
  char buf[64];
  std::time_t t = std::time(nullptr);
  std::tm tmv;
  localtime_r(&t, &tmv);
  std::strftime(buf, sizeof(buf), "%Y-%b-%d %H:%M:%S", &tmv);
  return std::string("[") + buf + "] ";
}

bool regular_nonempty(std::string const& p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0;
}

std::string file_name(std::string const& p) {
  const std::size_t s = p.find_last_of('/');
  return s == std::string::npos ? p : p.substr(s + 1);
}

std::string stem(std::string const& p) {
  std::string f = file_name(p);
  const std::size_t d = f.find_last_of('.');
  return (d == std::string::npos || d == 0) ? f : f.substr(0, d);
}

void usage_options(bool decompose) {
  std::cout << "Generic options:\n"
               "  -? [ --help ]                    show help message\n"
               "  -r [ --reference ] arg           fasta or wildtype ab1 file\n"
               "  -p [ --pratio ] arg (=0.33)      peak ratio to call base\n"
               "  -b [ --batch ] arg               manifest: trace<TAB>reference<TAB>outprefix per line\n"
               "  --threads arg (=0)               host threads of the --batch stages (0 = all usable cores)\n"
               "  -d [ --device ] arg (=0)         GPU ordinal, a list (0,1,2,3) or all: --batch manifests are cut into\n"
               "                                   contiguous blocks of traces, one per GPU\n";
  if (decompose)
    std::cout << "  -i [ --maxindel ] arg (=1000)    max. indel size in Sanger trace\n"
                 "  -v [ --callVariants ]            call variants in trace\n";
  std::cout << "\nAlignment options:\n"
               "  -g [ --gapopen ] arg (=-10)      gap open\n"
               "  -e [ --gapext ] arg (=-4)        gap extension\n"
               "  -m [ --match ] arg (=3)          match\n"
               "  -n [ --mismatch ] arg (=-5)      mismatch\n"
               "\nTrimming options:\n"
               "  -t [ --trim ] arg (=0)           trimming stringency [1:9], 0: use trimLeft and trimRight\n"
               "  -q [ --trimLeft ] arg (=50)      trim size left\n"
               "  -u [ --trimRight ] arg (=50)     trim size right\n"
               "\nOutput options:\n"
               "  -l [ --linelimit ] arg (=60)     alignment line length\n"
               "  -o [ --outprefix ] arg (=out)    output prefix\n\n";
}

void usage(const char* cmd) {
  std::cout << "Usage: tracy " << cmd << " [OPTIONS] -r genome.fa trace.ab1" << std::endl;
  usage_options(false);
}

// returns 0 ok, 1 show usage
int parse(int argc, char** argv, SageConfig& c) {
  static const std::map<std::string, char> longs = {
      {"help", '?'}, {"reference", 'r'}, {"pratio", 'p'}, {"batch", 'b'}, {"device", 'd'}, {"threads", 'T'}, {"gapopen", 'g'}, {"gapext", 'e'},
      {"match", 'm'}, {"mismatch", 'n'}, {"trim", 't'}, {"trimLeft", 'q'}, {"trimRight", 'u'}, {"linelimit", 'l'}, {"outprefix", 'o'},
      {"genome", 'r'}, {"maxindel", 'i'}, {"madc", 'c'}, {"qualCut", 'z'}, {"callVariants", 'v'}, {"annotate", 'a'}, {"kmer", 'k'},
      {"support", 's'}};
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    char opt = 0;
    std::string val;
    bool has_val = false;
    if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
      std::string name = a.substr(2);
      const std::size_t eq = name.find('=');
      if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
      auto it = longs.find(name);
      if (it == longs.end()) { std::cerr << "unrecognised option '" << a << "'" << std::endl; return 1; }
      opt = it->second;
    } else if (a.size() >= 2 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9')) {
      opt = a[1];
      if (a.size() > 2) { val = a.substr(2); has_val = true; }
    } else {
      c.ab = a;
      continue;
    }
    if (opt == '?') return 1;
    if (opt == 'v') { c.callvariants = true; continue; }
    if (!has_val) {
      if (i + 1 >= argc) { std::cerr << "the required argument for option '" << a << "' is missing" << std::endl; return 1; }
      val = argv[++i];
    }
    switch (opt) {
      case 'r': c.genome = val; break;
      case 'p': c.pratio = std::strtof(val.c_str(), nullptr); break;
      case 'b': c.batch = val; break;
      case 'T': c.threads = (uint32_t)std::atoi(val.c_str()); break;
      case 'd': c.devices = val; c.device = std::atoi(val.c_str()); break;
      case 'g': c.gapopen = std::atoi(val.c_str()); break;
      case 'e': c.gapext = std::atoi(val.c_str()); break;
      case 'm': c.match = std::atoi(val.c_str()); break;
      case 'n': c.mismatch = std::atoi(val.c_str()); break;
      case 't': c.trimStringency = std::strtof(val.c_str(), nullptr); break;
      case 'q': c.trimLeft = (uint16_t)std::atoi(val.c_str()); break;
      case 'u': c.trimRight = (uint16_t)std::atoi(val.c_str()); break;
      case 'l': c.linelimit = (uint16_t)std::atoi(val.c_str()); break;
      case 'o': c.outprefix = val; break;
      case 'i': c.maxindel = (uint16_t)std::atoi(val.c_str()); break;
      case 'c': c.madc = (uint16_t)std::atoi(val.c_str()); break;
      case 'z': c.qualCut = (uint16_t)std::atoi(val.c_str()); break;
      case 'a': c.annotate = val; break;
      case 'k': c.kmer = (uint16_t)std::atoi(val.c_str()); break;
      case 's': c.minKmerSupport = (uint16_t)std::atoi(val.c_str()); break;
      default: std::cerr << "unrecognised option '" << a << "'" << std::endl; return 1;
    }
  }
  if (c.maxindel < 1) c.maxindel = 1;
  return (c.ab.empty() && c.batch.empty()) ? 1 : 0;
}

bool load_trace(std::string const& path, Trace& tr) {
  detail::FileBytes f;  // read once: the format test and the reader look at the same bytes
  const int32_t ft = f.load(path) ? traceFormat(f) : -1;
  if (ft == 0) return readab(f, tr);
  if (ft == 1) return readscf(f, tr);
  std::cerr << "Unknown trace file type!" << std::endl;
  return false;
}

// k-mer tables of the indexed genomes named on the command line / in the manifest, built on first use
// (the reference loads the .fm9 written by `tracy index`; here the table is rebuilt in memory, seed.hpp)
const GenomeIndex* genome_index(SageConfig const& c, std::string const& path) {
  static std::map<std::string, GenomeIndex> cache;
  static std::mutex mtx;  // (prepare() runs on several threads in --batch mode)
  std::lock_guard<std::mutex> lock(mtx);
  auto it = cache.find(path);
  if (it != cache.end()) return &it->second;
  std::cout << stamp() << "Load FM-Index" << std::endl;
  GenomeIndex& g = cache[path];
  // an index file (`tracy_amd_cli index`; the reference loads genome.fa.gz.fm9, sage.h:203-207) is mapped; without one the table
  // is built in memory from the FASTA
  if (GenomeIndex::is_stale_index_file(path)) {
    std::cerr << "The index was written by an older version of this program: re-run `index` on the genome (" << path << ")." << std::endl;
    cache.erase(path);
    return nullptr;
  }
  if (GenomeIndex::is_stale_index_file(path + ".tidx"))
    std::cerr << "Warning: " << path << ".tidx was written by an older version and is ignored (re-run `index`); the k-mer table is rebuilt in memory." << std::endl;
  const std::string stored = GenomeIndex::is_index_file(path) ? path : (GenomeIndex::is_index_file(path + ".tidx") ? path + ".tidx" : std::string());
  if (!stored.empty()) {
    if (!g.open_index(stored)) {
      std::cerr << "Index file is corrupt: " << stored << std::endl;
      cache.erase(path);
      return nullptr;
    }
    if (g.k != c.kmer) {
      std::cerr << "The index was built for k-mers of " << g.k << " bases (-k " << c.kmer << " requested)." << std::endl;
      cache.erase(path);
      return nullptr;
    }
    return &g;
  }
  if (!g.load(path)) {
    std::cerr << "Couldn't recognize reference file format!" << std::endl;
    cache.erase(path);
    return nullptr;
  }
  g.build(c.kmer);
  return &g;
}

// host stages up to the device batch (sage.h:141-207, 222-231, 261-277); returns the CLI exit code
int prepare(SageConfig const& c, Job& j, bool decompose = false) {
  PhaseClock pc;
  if (!load_trace(j.trace_path, j.tr)) return -1;
  if (j.tr.basecallpos.empty()) {
    std::cerr << "Trace file lacks basecalls!" << std::endl;
    return -1;
  }
  pc.lap(CpuPhases::LOAD);
  basecall(j.tr, j.bc, c.pratio);
  pc.lap(CpuPhases::BASECALL);
  j.trimLeft = c.trimLeft;
  j.trimRight = c.trimRight;
  if (c.trimStringency >= 1) {
    uint32_t l = 0, r = 0;
    trimTrace(c.trimStringency, j.bc, l, r);
    j.trimLeft = (uint16_t)l;   // SageConfig stores the trims as uint16_t (sage.h:39-40)
    j.trimRight = (uint16_t)r;
  }
  if (j.trimLeft + j.trimRight >= j.bc.bcPos.size()) {
    std::cerr << "The sum of the left and right trim size is larger than the trace!" << std::endl;
    return -1;
  }
  traceTxtOut(j.outprefix + ".abif", j.bc, j.tr, j.trimLeft, j.trimRight);
  pc.lap(CpuPhases::ABIF_TXT);
  createProfile(j.tr, j.bc, j.full);
  pc.lap(CpuPhases::PROFILE);
  struct RefLap { PhaseClock& pc; ~RefLap() { pc.lap(CpuPhases::REFERENCE); } } ref_lap{pc};
  j.rs.filetype = genomeType(j.ref_path);
  if (j.rs.filetype == -1) {
    std::cerr << "Unknown reference file format!" << std::endl;
    return -1;
  }
  if (j.rs.filetype == 0) {
    // indexed genome (sage.h:217-221): anchor the trace by k-mer votes, take the window around the hit
    const GenomeIndex* idx = genome_index(c, j.ref_path);
    if (!idx) return -1;
    SeedConfig sc;
    sc.trimLeft = (uint16_t)j.trimLeft; sc.trimRight = (uint16_t)j.trimRight; sc.kmer = c.kmer; sc.minKmerSupport = c.minKmerSupport;
    sc.maxindel = c.maxindel;
    if (!getReferenceSlice(sc, *idx, j.bc.consensus, j.rs)) return -1;
    j.fasta = j.rs.refslice;  // already oriented
    j.slice_start = j.rs.pos;
    return 0;
  }
  if (j.rs.filetype == 1) {
    std::string name;
    if (!loadSingleFasta(j.ref_path, name, j.fasta)) return -1;
    if (j.fasta.size() > kMaxSingleFasta) {
      std::cerr << "Reference is larger than 50Kbp. Please use a smaller reference slice or an indexed genome!" << std::endl;
      return -1;
    }
    j.rs.chr = name;
  } else {
    Trace gtr;
    if (!load_trace(j.ref_path, gtr)) return -1;
    BaseCalls gbc;
    basecall(gtr, gbc, c.pratio);
    createProfile(gtr, gbc, j.wt_fwd);
    j.wt_primary = gbc.primary;
    j.rs.chr = "wildtype";
  }
  return 0;
}

// The GPUs a command runs on.  One ordinal: a plain context.  A list or "all": a device group (tracyhip_group_*: one context
// and one host thread per GPU); the batch pipelines go through the group, everything else through its first member.
struct Device {
  tracyhip_ctx* ctx = nullptr;
  tracyhip_group* group = nullptr;
  ~Device() {
    if (group) tracyhip_group_destroy(group);
    else if (ctx) tracyhip_destroy(ctx);
  }
  int open(std::string const& spec, uint32_t lanes) {
    std::vector<int> devs;
    bool all = (spec == "all");
    if (!all) {
      std::size_t pos = 0;
      while (pos <= spec.size()) {
        const std::size_t e = spec.find(',', pos);
        const std::string tok = spec.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) return -1;
        devs.push_back(std::atoi(tok.c_str()));
        if (e == std::string::npos) break;
        pos = e + 1;
      }
    }
    if (!all && devs.size() == 1) {
      if (tracyhip_create(devs[0], &ctx) != TRACYHIP_OK) return -1;
      tracyhip_set_lanes(ctx, lanes);
      return 0;
    }
    if (tracyhip_group_create(all ? nullptr : devs.data(), all ? 0 : (int)devs.size(), &group) != TRACYHIP_OK) return -1;
    tracyhip_group_set_lanes(group, lanes);
    ctx = tracyhip_group_context(group, 0);
    return 0;
  }
  int align_traces(const tracyhip_align_job* job, const tracyhip_params* prm, const tracyhip_align_result* res) {
    return group ? tracyhip_group_align_traces(group, job, prm, res) : tracyhip_align_traces(ctx, job, prm, TRACYHIP_MEM_HOST, res);
  }
  int decompose_traces(const tracyhip_decompose_job* job, const tracyhip_params* prm, const tracyhip_decompose_result* res) {
    return group ? tracyhip_group_decompose_traces(group, job, prm, res) : tracyhip_decompose_traces(ctx, job, prm, TRACYHIP_MEM_HOST, res);
  }
};

bool gpu_fail(const char* what) {
  std::cerr << "tracy_amd: " << what << ": " << tracyhip_last_error() << std::endl;
  return false;
}

// rows of gotoh(profile a1, a2) from its op string
bool alignment_rows(tracyhip_ctx* ctx, tracyhip_seqset const& s1, tracyhip_seqset const& s2, std::vector<uint8_t> const& ops,
                    std::vector<uint64_t> const& off, std::vector<uint32_t> const& len, std::vector<Job*> const& jobs, uint32_t nthreads) {
  tracyhip_pairs pr{};
  pr.npairs = (uint32_t)jobs.size();
  pr.a1 = s1;
  pr.a2 = s2;
  const std::size_t cap = ops.size() ? ops.size() : 1;
  std::unique_ptr<uint8_t[]> r0(new uint8_t[cap]), r1(new uint8_t[cap]);  // (written by the call: no zero fill of 100 MB)
  if (tracyhip_alignment_rows(ctx, &pr, TRACYHIP_MEM_HOST, ops.data(), off.data(), len.data(), r0.get(), r1.get()) != TRACYHIP_OK)
    return gpu_fail("alignment rows");
  for_each_index((uint32_t)jobs.size(), nthreads, [&](uint32_t i) {
    jobs[i]->rows.row0.assign(reinterpret_cast<char*>(r0.get()) + off[i], len[i]);
    jobs[i]->rows.row1.assign(reinterpret_cast<char*>(r1.get()) + off[i], len[i]);
  });
  return true;
}

// FASTA references: sage.h:233-260 + :311 for every job sharing one (trimLeft, trimRight)
bool align_fasta_group(Device& dev, tracyhip_params const& prm, std::vector<Job*> const& jobs, uint32_t nthreads) {
  PhaseClock pc;  // (on the calling thread: pack / device_call / unpack are wall seconds of the device stage)
  tracyhip_ctx* ctx = dev.ctx;
  const uint32_t nt = (uint32_t)jobs.size();
  // the batch as two packed payloads: offsets first, then every trace copied to its place by the host threads (a manifest of
  // 10 000 traces is 240 MB of profiles: grown by insert() it was copied several times over, by one thread)
  std::vector<uint64_t> poff(nt), roff(nt), ooff(nt);
  std::vector<uint32_t> plen(nt), rlen(nt);
  uint64_t ocap = 0, ptot = 0, rtot = 0;
  for (uint32_t i = 0; i < nt; ++i) {
    Job& j = *jobs[i];
    poff[i] = ptot;
    plen[i] = (uint32_t)j.full.cols;
    ptot += j.full.v.size();
    roff[i] = rtot;
    rlen[i] = (uint32_t)j.fasta.size();
    rtot += j.fasta.size();
    ooff[i] = ocap;
    ocap += (uint64_t)plen[i] + rlen[i];
  }
  std::unique_ptr<float[]> prof(new float[ptot ? ptot : 1]);
  std::unique_ptr<uint8_t[]> refs(new uint8_t[rtot ? rtot : 1]);
  for_each_index(nt, nthreads, [&](uint32_t i) {
    Job& j = *jobs[i];
    if (!j.full.v.empty()) std::memcpy(prof.get() + poff[i], j.full.v.data(), j.full.v.size() * sizeof(float));
    if (!j.fasta.empty()) std::memcpy(refs.get() + roff[i], j.fasta.data(), j.fasta.size());
  });
  tracyhip_align_job job{};
  job.ntraces = nt;
  job.profiles = tracyhip_seqset{TRACYHIP_SEQ_PROFILE, prof.get(), poff.data(), plen.data(), nt};
  job.refs = tracyhip_seqset{TRACYHIP_SEQ_CHAR, refs.get(), roff.data(), rlen.data(), nt};
  job.trim_left = jobs[0]->trimLeft;
  job.trim_right = jobs[0]->trimRight;
  const bool seeded = jobs[0]->rs.filetype == 0;  // groups never mix seeded and FASTA references
  std::vector<uint8_t> orient(nt);
  for (uint32_t i = 0; i < nt; ++i) orient[i] = jobs[i]->rs.forward ? 1 : 0;
  if (seeded) job.oriented = orient.data();
  job.strand_by_certificate = 1;  // gsFwd / gsRev are not part of any output file: only the decision (sage.h:247) matters here
  std::vector<int32_t> sf(nt), sr(nt), sfin(nt);
  std::vector<uint8_t> fwd(nt), ops(ocap ? ocap : 1);
  std::vector<uint32_t> sb(nt), sl(nt), rp(nt), olen(nt);
  tracyhip_align_result res{};
  res.score_fwd = sf.data(); res.score_rev = sr.data(); res.forward = fwd.data();
  res.slice_begin = sb.data(); res.slice_len = sl.data(); res.ref_pos = rp.data();
  res.score_final = sfin.data(); res.ops = ops.data(); res.ops_offset = ooff.data(); res.ops_len = olen.data();
  pc.lap(CpuPhases::PACK);
  if (dev.align_traces(&job, &prm, &res) != TRACYHIP_OK) return gpu_fail("align");
  pc.lap(CpuPhases::DEVICE_CALL);
  struct UnpackLap { PhaseClock& pc; ~UnpackLap() { pc.lap(CpuPhases::UNPACK); } } unpack_lap{pc};
  // the reference slices the final alignment ran against (trimReferenceSlice, fmindex.h:429-463)
  std::vector<uint64_t> soff(nt);
  uint64_t stot = 0;
  for (uint32_t i = 0; i < nt; ++i) { soff[i] = stot; stot += sl[i]; }
  std::unique_ptr<uint8_t[]> slices(new uint8_t[stot ? stot : 1]);
  for_each_index(nt, nthreads, [&](uint32_t i) {
    Job& j = *jobs[i];
    j.rs.forward = fwd[i] != 0;
    std::string oriented = j.fasta;
    if (!seeded && !j.rs.forward) reverseComplement(oriented);
    j.rs.refslice = oriented.substr(sb[i], sl[i]);
    j.rs.pos = j.slice_start + rp[i];
    j.score = sfin[i];
    if (!j.rs.refslice.empty()) std::memcpy(slices.get() + soff[i], j.rs.refslice.data(), j.rs.refslice.size());
  });
  tracyhip_seqset s2{TRACYHIP_SEQ_CHAR, slices.get(), soff.data(), sl.data(), nt};
  return alignment_rows(ctx, job.profiles, s2, ops, ooff, olen, jobs, nthreads);
}

// wildtype-trace references: sage.h:261-301 + :311 (profile x profile)
bool align_wildtype_group(tracyhip_ctx* ctx, tracyhip_params const& prm, std::vector<Job*> const& jobs) {
  const uint32_t nt = (uint32_t)jobs.size();
  std::vector<Profile> trimmed(nt), wrev(nt);
  std::vector<float> ptrim, pfull, pref;  // pref: forward then reverse-complement wildtype profile per job
  std::vector<uint64_t> toff(nt), foff(nt), woff(2 * nt), ooff(nt);
  std::vector<uint32_t> tlen(nt), flen(nt), wlen(2 * nt), idx1(2 * nt), idx2(2 * nt);
  for (uint32_t i = 0; i < nt; ++i) {
    Job& j = *jobs[i];
    createProfile(j.tr, j.bc, trimmed[i], (int32_t)j.trimLeft, (int32_t)j.trimRight);
    reverseComplementProfile(j.wt_fwd, wrev[i]);
    toff[i] = ptrim.size(); tlen[i] = (uint32_t)trimmed[i].cols;
    ptrim.insert(ptrim.end(), trimmed[i].v.begin(), trimmed[i].v.end());
    foff[i] = pfull.size(); flen[i] = (uint32_t)j.full.cols;
    pfull.insert(pfull.end(), j.full.v.begin(), j.full.v.end());
    for (int r = 0; r < 2; ++r) {
      Profile const& p = r ? wrev[i] : j.wt_fwd;
      woff[2 * i + r] = pref.size(); wlen[2 * i + r] = (uint32_t)p.cols;
      pref.insert(pref.end(), p.v.begin(), p.v.end());
      idx1[2 * i + r] = i; idx2[2 * i + r] = 2 * i + r;
    }
  }
  tracyhip_pairs sp{};
  sp.npairs = 2 * nt;
  sp.a1 = tracyhip_seqset{TRACYHIP_SEQ_PROFILE, ptrim.data(), toff.data(), tlen.data(), nt};
  sp.a2 = tracyhip_seqset{TRACYHIP_SEQ_PROFILE, pref.data(), woff.data(), wlen.data(), 2 * nt};
  sp.a1_index = idx1.data(); sp.a2_index = idx2.data();
  std::vector<int32_t> gs(2 * nt);
  if (tracyhip_gotoh_score(ctx, &sp, &prm, TRACYHIP_MEM_HOST, gs.data()) != TRACYHIP_OK) return gpu_fail("orientation scores");
  std::vector<uint32_t> pick(nt), olen(nt);
  uint64_t ocap = 0;
  for (uint32_t i = 0; i < nt; ++i) {
    Job& j = *jobs[i];
    j.rs.forward = gs[2 * i] > gs[2 * i + 1];
    j.rs.refslice = j.wt_primary;
    if (!j.rs.forward) reverseComplement(j.rs.refslice);
    j.rs.pos = 0;
    pick[i] = 2 * i + (j.rs.forward ? 0 : 1);
    ooff[i] = ocap;
    ocap += (uint64_t)flen[i] + wlen[pick[i]];
  }
  tracyhip_pairs fp{};
  fp.npairs = nt;
  fp.a1 = tracyhip_seqset{TRACYHIP_SEQ_PROFILE, pfull.data(), foff.data(), flen.data(), nt};
  fp.a2 = sp.a2;
  fp.a2_index = pick.data();
  std::vector<int32_t> sc(nt);
  std::vector<uint8_t> ops(ocap ? ocap : 1);
  if (tracyhip_gotoh_align(ctx, &fp, &prm, TRACYHIP_MEM_HOST, sc.data(), ops.data(), ooff.data(), olen.data()) != TRACYHIP_OK)
    return gpu_fail("final alignment");
  tracyhip_pairs rp = fp;
  std::vector<uint8_t> r0(ops.size()), r1(ops.size());
  if (tracyhip_alignment_rows(ctx, &rp, TRACYHIP_MEM_HOST, ops.data(), ooff.data(), olen.data(), r0.data(), r1.data()) != TRACYHIP_OK)
    return gpu_fail("alignment rows");
  for (uint32_t i = 0; i < nt; ++i) {
    jobs[i]->score = sc[i];
    jobs[i]->rows.row0.assign(reinterpret_cast<char*>(r0.data()) + ooff[i], olen[i]);
    jobs[i]->rows.row1.assign(reinterpret_cast<char*>(r1.data()) + ooff[i], olen[i]);
  }
  return true;
}

// sage.h:313-345
void write_outputs(SageConfig const& c, Job const& j) {
  PhaseClock pc;
  PaddedTrace padded;
  alignmentTracePadding(j.rows.row0, j.tr, j.bc, padded);
  pc.lap(CpuPhases::PAD);
  {
    TextBuf f(2 * j.rows.row0.size() + 512);
    alignFastaOut(f, stem(j.trace_path), j.rs, j.rows);
    f.write(j.outprefix + ".align.fa");
  }
  plotAlignment(j.outprefix + ".txt", j.rows, j.rs, j.score, c.linelimit);
  pc.lap(CpuPhases::SMALL_FILES);
  traceAlignJsonOut(j.outprefix + ".json", padded, j.rs, j.rows);
  pc.lap(CpuPhases::JSON);
}

// one job from the command line, or one per manifest line; returns the CLI exit code
int collect_jobs(SageConfig const& c, std::vector<Job>& jobs) {
  if (!c.batch.empty()) {
    std::ifstream mf(c.batch.c_str());
    if (!mf) {
      std::cerr << "Manifest is missing: " << c.batch << std::endl;
      return 1;
    }
    std::string line;
    while (std::getline(mf, line)) {
      if (line.empty() || line[0] == '#') continue;
      std::istringstream ss(line);
      Job j;
      if (!std::getline(ss, j.trace_path, '\t') || !std::getline(ss, j.ref_path, '\t') || !std::getline(ss, j.outprefix, '\t')) {
        std::cerr << "Malformed manifest line: " << line << std::endl;
        return 1;
      }
      jobs.push_back(std::move(j));
    }
  } else {
    Job j;
    j.trace_path = c.ab;
    j.ref_path = c.genome;
    j.outprefix = c.outprefix;
    jobs.push_back(std::move(j));
  }
  return 0;
}

void echo_command(int argc, char** argv) {
  std::cout << stamp() << "tracy ";
  for (int i = 0; i < argc; ++i) std::cout << argv[i] << ' ';
  std::cout << std::endl;
}

int align_main(int argc, char** argv) {
  SageConfig c;
  if (parse(argc, argv, c)) {
    usage(argv[0]);
    return -1;
  }
  if (c.trimStringency > 9) c.trimStringency = 9;
  std::vector<Job> jobs;
  const bool batch = !c.batch.empty();
  if (int rc = collect_jobs(c, jobs)) return rc;
  for (Job const& j : jobs) {
    if (!regular_nonempty(j.ref_path)) {
      std::cerr << "Reference file is missing: " << file_name(j.ref_path) << std::endl;
      return 1;
    }
    if (!regular_nonempty(j.trace_path)) {
      std::cerr << "Input trace file is missing: " << file_name(j.trace_path) << std::endl;
      return 1;
    }
  }
  echo_command(argc, argv);

  std::cout << stamp() << "Load ab1 file" << std::endl;
  std::atomic<int> failed(0);
  const uint32_t nthreads = batch ? (c.threads ? c.threads : usable_cores()) : 1;
  StageTimes times;
  int fatal = 0;  // a single trace that cannot be prepared ends the command with its code (the reference's behaviour)
  std::vector<int> rcs(jobs.size(), 0);
  auto prep = [&](uint32_t lo, uint32_t hi) {
    Stopwatch sw;
    for_each_index(hi - lo, nthreads, [&](uint32_t i) { rcs[lo + i] = prepare(c, jobs[lo + i]); });
    for (uint32_t i = lo; i < hi; ++i) {
      if (rcs[i] != 0) {
        if (!batch) continue;
        std::cerr << "skipping " << jobs[i].trace_path << std::endl;
        ++failed;
      } else {
        jobs[i].ok = true;
      }
    }
    times.add("read_basecall_profile_s", sw.seconds());
  };
  Device dev;
  bool dev_open = false, said = false;
  // the context is created (HIP start-up, ~0.1 s) while the first block is read
  std::future<int> dev_ready = std::async(std::launch::async, [&]() {
    Stopwatch sw;
    const int rc = dev.open(c.devices, 2);  // two chunks of a block in flight per GPU (small batches run on one lane)
    times.add("gpu_init_s", sw.seconds());
    return rc;
  });
  tracyhip_params prm{c.match, c.mismatch, c.gapopen, c.gapext, 1, 0};  // AlignConfig<true,false>, sage.h:165
  auto device = [&](uint32_t lo, uint32_t hi) -> bool {
    if (!batch && rcs[lo] != 0) { fatal = rcs[lo]; return false; }
    if (!said) std::cout << stamp() << "Find reference match" << std::endl;
    if (!dev_open) {
      if (dev_ready.get() != 0) {
        gpu_fail("no usable GPU");
        fatal = -1;
        return false;
      }
      dev_open = true;
    }
    Stopwatch sw;
    std::map<std::pair<uint32_t, uint32_t>, std::vector<Job*>> fasta_groups, seeded_groups;
    std::vector<Job*> wildtype;
    for (uint32_t i = lo; i < hi; ++i) {
      Job& j = jobs[i];
      if (!j.ok) continue;
      if (j.rs.filetype == 1) fasta_groups[std::make_pair(j.trimLeft, j.trimRight)].push_back(&j);
      else if (j.rs.filetype == 0) seeded_groups[std::make_pair(j.trimLeft, j.trimRight)].push_back(&j);
      else wildtype.push_back(&j);
    }
    if (!said) std::cout << stamp() << "Alignment" << std::endl;
    said = true;
    fatal = -1;
    for (auto& g : fasta_groups)
      if (!align_fasta_group(dev, prm, g.second, nthreads)) return false;
    for (auto& g : seeded_groups)
      if (!align_fasta_group(dev, prm, g.second, nthreads)) return false;
    if (!wildtype.empty() && !align_wildtype_group(dev.ctx, prm, wildtype)) return false;
    fatal = 0;
    times.add("device_s", sw.seconds());
    return true;
  };
  bool said_out = false;
  auto write = [&](uint32_t lo, uint32_t hi) {
    Stopwatch sw;
    if (!said_out) std::cout << stamp() << "Output" << std::endl;
    said_out = true;
    for_each_index(hi - lo, nthreads, [&](uint32_t i) {
      Job& j = jobs[lo + i];
      if (j.ok) write_outputs(c, j);
      if (batch) { Job done; done.trace_path.swap(j.trace_path); done.ok = j.ok; j = std::move(done); }  // the block's traces are released here
    });
    times.add("writers_s", sw.seconds());
  };
  if (!run_blocks((uint32_t)jobs.size(), batch ? block_size() : (uint32_t)jobs.size(), prep, device, write)) return fatal ? fatal : -1;
  times.report((uint32_t)jobs.size(), nthreads);
  std::cout << stamp() << "Done." << std::endl;
  const int rc_out = (failed || TextBuf::write_errors()) ? 2 : 0;  // (a file that could not be written counts like a trace that failed)
  if (batch) end_process(rc_out);
  return rc_out;
}


// ---- `tracy decompose` (indigo.h:42-455) ----------------------------------------------------------------

// char x char alignments through the C ABI: scores, rows
bool align_strings(tracyhip_ctx* ctx, tracyhip_params const& prm, std::vector<std::string> const& a1, std::vector<std::string> const& a2,
                   std::vector<int32_t>& scores, std::vector<AlignRows>& rows) {
  const uint32_t np = (uint32_t)a1.size();
  scores.assign(np, 0);
  rows.assign(np, AlignRows());
  if (np == 0) return true;
  std::vector<uint8_t> d1, d2;
  std::vector<uint64_t> o1(np), o2(np), oo(np);
  std::vector<uint32_t> l1(np), l2(np), olen(np);
  uint64_t cap = 0;
  for (uint32_t i = 0; i < np; ++i) {
    o1[i] = d1.size(); l1[i] = (uint32_t)a1[i].size(); d1.insert(d1.end(), a1[i].begin(), a1[i].end());
    o2[i] = d2.size(); l2[i] = (uint32_t)a2[i].size(); d2.insert(d2.end(), a2[i].begin(), a2[i].end());
    oo[i] = cap; cap += (uint64_t)l1[i] + l2[i];
  }
  d1.push_back(0); d2.push_back(0);
  tracyhip_pairs pr{};
  pr.npairs = np;
  pr.a1 = tracyhip_seqset{TRACYHIP_SEQ_CHAR, d1.data(), o1.data(), l1.data(), np};
  pr.a2 = tracyhip_seqset{TRACYHIP_SEQ_CHAR, d2.data(), o2.data(), l2.data(), np};
  std::vector<uint8_t> ops(cap ? cap : 1), r0(cap ? cap : 1), r1(cap ? cap : 1);
  if (tracyhip_gotoh_align(ctx, &pr, &prm, TRACYHIP_MEM_HOST, scores.data(), ops.data(), oo.data(), olen.data()) != TRACYHIP_OK)
    return gpu_fail("alignment");
  if (tracyhip_alignment_rows(ctx, &pr, TRACYHIP_MEM_HOST, ops.data(), oo.data(), olen.data(), r0.data(), r1.data()) != TRACYHIP_OK)
    return gpu_fail("alignment rows");
  for (uint32_t i = 0; i < np; ++i) {
    rows[i].row0.assign(reinterpret_cast<char*>(r0.data()) + oo[i], olen[i]);
    rows[i].row1.assign(reinterpret_cast<char*>(r1.data()) + oo[i], olen[i]);
  }
  return true;
}

// rows of alignments whose op strings came back from tracyhip_decompose_traces
bool rows_from_ops(tracyhip_ctx* ctx, std::vector<std::string> const& a1, std::vector<std::string> const& a2, std::vector<uint8_t> const& ops,
                   std::vector<uint64_t> const& off, std::vector<uint32_t> const& len, std::vector<AlignRows>& rows, uint32_t nthreads = 1) {
  const uint32_t np = (uint32_t)a1.size();
  rows.assign(np, AlignRows());
  // packed payloads: offsets first, then every string copied to its place by the host threads
  std::vector<uint64_t> o1(np), o2(np);
  std::vector<uint32_t> l1(np), l2(np);
  uint64_t t1 = 0, t2 = 0;
  for (uint32_t i = 0; i < np; ++i) {
    o1[i] = t1; l1[i] = (uint32_t)a1[i].size(); t1 += l1[i];
    o2[i] = t2; l2[i] = (uint32_t)a2[i].size(); t2 += l2[i];
  }
  std::unique_ptr<uint8_t[]> d1(new uint8_t[t1 + 1]), d2(new uint8_t[t2 + 1]);
  for_each_index(np, nthreads, [&](uint32_t i) {
    if (l1[i]) std::memcpy(d1.get() + o1[i], a1[i].data(), l1[i]);
    if (l2[i]) std::memcpy(d2.get() + o2[i], a2[i].data(), l2[i]);
  });
  d1[t1] = 0; d2[t2] = 0;
  tracyhip_pairs pr{};
  pr.npairs = np;
  pr.a1 = tracyhip_seqset{TRACYHIP_SEQ_CHAR, d1.get(), o1.data(), l1.data(), np};
  pr.a2 = tracyhip_seqset{TRACYHIP_SEQ_CHAR, d2.get(), o2.data(), l2.data(), np};
  const std::size_t cap = ops.size() ? ops.size() : 1;
  std::unique_ptr<uint8_t[]> r0(new uint8_t[cap]), r1(new uint8_t[cap]);  // (written by the call: no zero fill)
  if (tracyhip_alignment_rows(ctx, &pr, TRACYHIP_MEM_HOST, ops.data(), off.data(), len.data(), r0.get(), r1.get()) != TRACYHIP_OK)
    return gpu_fail("alignment rows");
  for_each_index(np, nthreads, [&](uint32_t i) {
    rows[i].row0.assign(reinterpret_cast<char*>(r0.get()) + off[i], len[i]);
    rows[i].row1.assign(reinterpret_cast<char*>(r1.get()) + off[i], len[i]);
  });
  return true;
}

// wildtype-trace references (indigo.h:249-289): strand by gotohScore(trimmed trace, wildtype profile / its reverse
// complement); leaves the oriented profile in j.wt_fwd and the oriented primary calls in j.fasta
bool orient_wildtype(tracyhip_ctx* ctx, tracyhip_params const& prm, std::vector<Job*> const& jobs) {
  const uint32_t nt = (uint32_t)jobs.size();
  std::vector<Profile> trimmed(nt), wrev(nt);
  std::vector<float> ptrim, pref;
  std::vector<uint64_t> toff(nt), woff(2 * nt);
  std::vector<uint32_t> tlen(nt), wlen(2 * nt), idx1(2 * nt), idx2(2 * nt);
  for (uint32_t i = 0; i < nt; ++i) {
    Job& j = *jobs[i];
    createProfile(j.tr, j.bc, trimmed[i], (int32_t)j.trimLeft, (int32_t)j.trimRight);
    reverseComplementProfile(j.wt_fwd, wrev[i]);
    toff[i] = ptrim.size(); tlen[i] = (uint32_t)trimmed[i].cols;
    ptrim.insert(ptrim.end(), trimmed[i].v.begin(), trimmed[i].v.end());
    for (int r = 0; r < 2; ++r) {
      Profile const& p = r ? wrev[i] : j.wt_fwd;
      woff[2 * i + r] = pref.size(); wlen[2 * i + r] = (uint32_t)p.cols;
      pref.insert(pref.end(), p.v.begin(), p.v.end());
      idx1[2 * i + r] = i; idx2[2 * i + r] = 2 * i + r;
    }
  }
  tracyhip_pairs sp{};
  sp.npairs = 2 * nt;
  sp.a1 = tracyhip_seqset{TRACYHIP_SEQ_PROFILE, ptrim.data(), toff.data(), tlen.data(), nt};
  sp.a2 = tracyhip_seqset{TRACYHIP_SEQ_PROFILE, pref.data(), woff.data(), wlen.data(), 2 * nt};
  sp.a1_index = idx1.data(); sp.a2_index = idx2.data();
  std::vector<int32_t> gs(2 * nt);
  if (tracyhip_gotoh_score(ctx, &sp, &prm, TRACYHIP_MEM_HOST, gs.data()) != TRACYHIP_OK) return gpu_fail("orientation scores");
  for (uint32_t i = 0; i < nt; ++i) {
    Job& j = *jobs[i];
    j.rs.forward = gs[2 * i] > gs[2 * i + 1];
    j.fasta = j.wt_primary;
    if (!j.rs.forward) { reverseComplement(j.fasta); j.wt_fwd = wrev[i]; }
  }
  return true;
}

// indigo.h:190-388 for every job sharing one (trimLeft, trimRight): the whole chain runs on the device
// One group of a block (same reference kind and trims) on its way through tracyhip_decompose_traces: pack() lays the batch out as
// packed payloads and sizes the result arrays -- host work, done by the stage that prepares the block (for wildtype references by
// the device stage, once the device has oriented them); run() is the call, the results back into the jobs and the alignment rows.
struct DecomposeGroup {
  std::vector<Job*> jobs;
  uint32_t nt = 0;
  bool wildtype = false, seeded = false, packed = false;
  std::vector<uint64_t> poff, roff, boff, dcpoff, ooff[3], wpoff;
  std::vector<uint32_t> plen, rlen, blen, sb[2], sl[2], rp[2], olen[3];
  uint32_t dcap = 0;
  uint64_t ocap[3] = {0, 0, 0}, btot = 0;
  struct Packed {
    std::unique_ptr<float[]> prof;
    std::unique_ptr<uint8_t[]> refs, pri, sec;
    std::unique_ptr<int32_t[]> peaks;  // tracyhip_basecalls::peaks: the four channels at every basecall's peak position (all the device reads of a chromatogram)
    float* pdata() { return prof.get(); }
  };
  Packed pk;
  uint8_t *pri_p = nullptr, *sec_p = nullptr;
  tracyhip_decompose_job job{};
  tracyhip_decompose_result res{};
  std::vector<uint8_t> orient, fwd, sd, ops[3];
  std::vector<float> wprof;
  std::vector<tracyhip_breakpoint> bp;
  std::vector<int32_t> status, sf, sr, strim, dci, dce, sc[3];
  std::vector<tracyhip_decomp_status> dst;
  std::vector<double> fr;

  explicit DecomposeGroup(std::vector<Job*> js) : jobs(std::move(js)), nt((uint32_t)jobs.size()), wildtype(jobs[0]->rs.filetype == 2) {}  // (groups never mix reference kinds)

  void pack(SageConfig const& c, uint32_t nthreads) {
    PhaseClock pc;
    // the batch as packed payloads: offsets first, then every trace copied to its place by the host threads.  Of a chromatogram the
    // device reads the samples at the basecalls' peak positions only (generateSecondaryDecomposed, allelicFraction): the block carries the
    // peak table -- 16 bytes per basecall -- instead of 1.9 GB of signal per 10 000 traces
    for (auto* v : {&poff, &roff, &boff, &dcpoff}) v->assign(nt, 0);
    for (auto* v : {&plen, &rlen, &blen}) v->assign(nt, 0);
    dcap = 2u * c.maxindel + 2;
    for (int k = 0; k < 3; ++k) ocap[k] = 0;
    for (int k = 0; k < 3; ++k) ooff[k].resize(nt);
    uint64_t ptot = 0, rtot = 0;
    btot = 0;
    for (uint32_t i = 0; i < nt; ++i) {
      Job& j = *jobs[i];
      poff[i] = ptot; plen[i] = (uint32_t)j.full.cols; ptot += j.full.v.size();
      roff[i] = rtot; rlen[i] = (uint32_t)j.fasta.size(); rtot += j.fasta.size();
      boff[i] = btot; blen[i] = (uint32_t)j.bc.bcPos.size(); btot += blen[i];
      dcpoff[i] = (uint64_t)i * dcap;
      for (int k = 0; k < 3; ++k) {
        ooff[k][i] = ocap[k];
        ocap[k] += (uint64_t)blen[i] + (k < 2 ? rlen[i] : blen[i]);
      }
    }
    pk.prof.reset(new float[ptot ? ptot : 1]);
    pk.refs.reset(new uint8_t[rtot ? rtot : 1]);
    pk.pri.reset(new uint8_t[btot ? btot : 1]);
    pk.sec.reset(new uint8_t[btot ? btot : 1]);
    pk.peaks.reset(new int32_t[btot ? 4 * btot : 4]);
    for_each_index(nt, nthreads, [&](uint32_t i) {
      Job& j = *jobs[i];
      if (!j.full.v.empty()) std::memcpy(pk.prof.get() + poff[i], j.full.v.data(), j.full.v.size() * sizeof(float));
      if (!j.fasta.empty()) std::memcpy(pk.refs.get() + roff[i], j.fasta.data(), j.fasta.size());
      // (the per-base arrays share bc_offset: shorter ones are an input error the library reports)
      const std::size_t nb = blen[i];
      if (nb) {
        int32_t* pt = pk.peaks.get() + 4 * boff[i];
        for (std::size_t b = 0; b < nb; ++b) {
          const std::size_t at = (std::size_t)j.bc.bcPos[b];
          for (int k = 0; k < 4; ++k) pt[4 * b + k] = at < j.tr.traceACGT[k].size() ? j.tr.traceACGT[k][at] : 0;  // (a sample behind a channel's end reads as zero)
        }
        std::memcpy(pk.pri.get() + boff[i], j.bc.primary.data(), std::min<std::size_t>(nb, j.bc.primary.size()));
        std::memcpy(pk.sec.get() + boff[i], j.bc.secondary.data(), std::min<std::size_t>(nb, j.bc.secondary.size()));
      }
    });
    float* const prof_p = pk.prof.get();
    uint8_t* const refs_p = pk.refs.get();
    pri_p = pk.pri.get();
    sec_p = pk.sec.get();
    job = tracyhip_decompose_job{};
    job.ntraces = nt;
    job.profiles = tracyhip_seqset{TRACYHIP_SEQ_PROFILE, prof_p, poff.data(), plen.data(), nt};
    job.bc = tracyhip_basecalls{nt, nullptr, nullptr, nullptr, nullptr, pri_p, sec_p, boff.data(), blen.data(), pk.peaks.get()};
    job.refs = tracyhip_seqset{TRACYHIP_SEQ_CHAR, refs_p, roff.data(), rlen.data(), nt};
    job.dprm = tracyhip_decomp_params{(int32_t)jobs[0]->trimLeft, (int32_t)jobs[0]->trimRight, (int32_t)c.maxindel, (int32_t)c.madc};
    job.strand_by_certificate = 1;  // the orientation scores are not written anywhere (indigo.h:235-247 keeps only rs.forward)
    seeded = jobs[0]->rs.filetype == 0 || wildtype;  // the reference arrives oriented
    orient.assign(nt, 0);
    for (uint32_t i = 0; i < nt; ++i) orient[i] = jobs[i]->rs.forward ? 1 : 0;
    if (seeded) job.oriented = orient.data();
    wprof.clear();
    wpoff.assign(nt, 0);
    if (wildtype) {
      for (uint32_t i = 0; i < nt; ++i) {
        wpoff[i] = wprof.size();
        wprof.insert(wprof.end(), jobs[i]->wt_fwd.v.begin(), jobs[i]->wt_fwd.v.end());
      }
      job.ref_profiles = tracyhip_seqset{TRACYHIP_SEQ_PROFILE, wprof.data(), wpoff.data(), rlen.data(), nt};
    }
    bp.assign(nt, tracyhip_breakpoint{});
    for (auto* v : {&status, &sf, &sr, &strim}) v->assign(nt, 0);
    dci.assign((size_t)nt * dcap, 0); dce.assign((size_t)nt * dcap, 0);
    fwd.assign(nt, 0); sd.assign(btot ? btot : 1, 0);
    dst.assign(nt, tracyhip_decomp_status{});
    fr.assign(2 * (size_t)nt, 0.0);
    res = tracyhip_decompose_result{};
    res.bp = bp.data(); res.status = status.data(); res.score_fwd = sf.data(); res.score_rev = sr.data(); res.forward = fwd.data();
    res.score_trim = strim.data(); res.dcp_indel = dci.data(); res.dcp_err = dce.data(); res.dcp_offset = dcpoff.data();
    res.dstatus = dst.data(); res.secdecomp = sd.data(); res.fractions = fr.data();
    for (int k = 0; k < 3; ++k) {
      if (k < 2) {
        sb[k].resize(nt); sl[k].resize(nt); rp[k].resize(nt);
        res.slice_begin[k] = sb[k].data(); res.slice_len[k] = sl[k].data(); res.ref_pos[k] = rp[k].data();
      }
      sc[k].resize(nt); olen[k].resize(nt); ops[k].resize(ocap[k] ? ocap[k] : 1);
      res.score[k] = sc[k].data(); res.ops[k] = ops[k].data(); res.ops_offset[k] = ooff[k].data(); res.ops_len[k] = olen[k].data();
    }
    packed = true;
    pc.lap(CpuPhases::PACK);
  }

  bool run(Device& dev, SageConfig const& c, tracyhip_params const& prm, uint32_t nthreads) {
    PhaseClock pc;  // (on the calling thread: device_call / unpack are wall seconds of the device stage)
    tracyhip_ctx* ctx = dev.ctx;
    if (wildtype && !orient_wildtype(ctx, prm, jobs)) return false;
    if (!packed) pack(c, nthreads);
    pc.lap(CpuPhases::PACK);
    if (dev.decompose_traces(&job, &prm, &res) != TRACYHIP_OK) return gpu_fail("decompose");
    pc.lap(CpuPhases::DEVICE_CALL);
    struct UnpackLap { PhaseClock& pc; ~UnpackLap() { pc.lap(CpuPhases::UNPACK); } } unpack_lap{pc};

    std::vector<std::string> a1[3], a2[3];
    for (int k = 0; k < 3; ++k) { a1[k].resize(nt); a2[k].resize(nt); }
    for_each_index(nt, nthreads, [&](uint32_t i) {
      Job& j = *jobs[i];
      j.status = status[i];
      j.dstatus = dst[i];
      j.primary.assign(reinterpret_cast<char*>(pri_p) + boff[i], blen[i]);
      j.secondary.assign(reinterpret_cast<char*>(sec_p) + boff[i], blen[i]);
      j.secdecomp.assign(reinterpret_cast<char*>(sd.data()) + boff[i], blen[i]);
      j.rs.forward = fwd[i] != 0;
      j.rs.refslice = j.fasta;
      if (!seeded && !j.rs.forward) reverseComplement(j.rs.refslice);
      j.rs.pos = j.slice_start;
      AlleleReport& r = j.rep;
      r.bp.indelshift = bp[i].indelshift != 0;
      r.bp.traceleft = bp[i].traceleft != 0;
      r.bp.breakpoint = bp[i].breakpoint;
      r.bp.bestDiff = bp[i].best_diff;
      r.a1a2 = std::make_pair(fr[2 * i], fr[2 * i + 1]);
      r.dcp.clear();
      for (uint32_t k = 0; k < dst[i].dcp_n; ++k) r.dcp.emplace_back(dci[dcpoff[i] + k], dce[dcpoff[i] + k]);
      const std::string p_t = trimmedSeq(j.primary, j.trimLeft, j.trimRight), s_t = trimmedSeq(j.secdecomp, j.trimLeft, j.trimRight);
      ReferenceSlice* slot[2] = {&r.rs1, &r.rs2};
      for (int k = 0; k < 2; ++k) {
        *slot[k] = j.rs;
        const bool usable = status[i] == 0;
        slot[k]->refslice = usable ? j.rs.refslice.substr(sb[k][i], sl[k][i]) : std::string();
        slot[k]->pos = usable ? j.slice_start + rp[k][i] : 0;
        a1[k][i] = k == 0 ? p_t : s_t;
        a2[k][i] = slot[k]->refslice;
      }
      a1[2][i] = p_t;
      a2[2][i] = s_t;
      r.a1Score = sc[0][i]; r.a2Score = sc[1][i]; r.a3Score = sc[2][i];
      if (status[i] != 0)
        for (int k = 0; k < 3; ++k) olen[k][i] = 0;
    });
    for (int k = 0; k < 3; ++k) {
      std::vector<AlignRows> rows;
      if (!rows_from_ops(ctx, a1[k], a2[k], ops[k], ooff[k], olen[k], rows, nthreads)) return false;
      for_each_index(nt, nthreads, [&](uint32_t i) { (k == 0 ? jobs[i]->rep.align1 : k == 1 ? jobs[i]->rep.align2 : jobs[i]->rep.align3) = std::move(rows[i]); });
    }
    return true;
  }
};

// variants of both alleles (indigo.h:393-421); reverse-strand traces are re-aligned as reverse complements
bool call_variants(tracyhip_ctx* ctx, tracyhip_params const& prm, std::vector<Job*> const& jobs, uint32_t nthreads = 1) {
  // traces on the reverse strand are called on re-aligned reverse complements (two pairs each, one device batch); the rest
  // straight from their allele alignments -- every trace on its own, on the host threads
  std::vector<uint32_t> slot(jobs.size(), 0);
  uint32_t nrev = 0;
  for (std::size_t i = 0; i < jobs.size(); ++i)
    if (!jobs[i]->rs.forward) slot[i] = nrev++;
  std::vector<std::string> s1(2 * (std::size_t)nrev), s2(2 * (std::size_t)nrev);
  std::vector<ReferenceSlice> rev(2 * (std::size_t)nrev);
  for_each_index((uint32_t)jobs.size(), nthreads, [&](uint32_t i) {
    Job& j = *jobs[i];
    AlleleReport& r = j.rep;
    if (j.rs.forward) {
      callVariants(r.align1, r.rs1, r.var);
      callVariants(r.align2, r.rs2, r.var);
      return;
    }
    const std::string seq[2] = {trimmedSeq(j.primary, j.trimLeft, j.trimRight), trimmedSeq(j.secdecomp, j.trimLeft, j.trimRight)};
    ReferenceSlice const* rs[2] = {&r.rs1, &r.rs2};
    for (int k = 0; k < 2; ++k) {
      const std::size_t at = 2 * (std::size_t)slot[i] + k;
      s1[at] = seq[k];
      reverseComplement(s1[at]);
      reverseReferenceSlice(*rs[k], rev[at]);
      s2[at] = rev[at].refslice;
    }
  });
  std::vector<int32_t> scores;
  std::vector<AlignRows> rows;
  if (!align_strings(ctx, prm, s1, s2, scores, rows)) return false;
  for_each_index((uint32_t)jobs.size(), nthreads, [&](uint32_t i) {
    Job& j = *jobs[i];
    if (!j.rs.forward)
      for (int k = 0; k < 2; ++k) callVariants(rows[2 * (std::size_t)slot[i] + k], rev[2 * (std::size_t)slot[i] + k], j.rep.var);
    std::sort(j.rep.var.begin(), j.rep.var.end());
  });
  return true;
}

// indigo.h:340-343, 359-387, 389-394, 436-442
void write_decompose_outputs(SageConfig const& c, Job& j) {
  PhaseClock pc;
  AlleleReport& r = j.rep;
  {
    TextBuf f(4096);
    writeDecomposition(f, r.dcp);
    f.write(j.outprefix + ".decomp");
  }
  ReferenceSlice secrs;
  secrs.refslice = trimmedSeq(j.secdecomp, j.trimLeft, j.trimRight);
  secrs.forward = true;
  secrs.pos = 0;
  secrs.chr = "Alt2";
  const int32_t score[3] = {r.a1Score, r.a2Score, r.a3Score};
  AlignRows const* al[3] = {&r.align1, &r.align2, &r.align3};
  ReferenceSlice const* rs[3] = {&r.rs1, &r.rs2, &secrs};
  for (int k = 0; k < 3; ++k) {
    TextBuf f;
    plotAlignment(f, *al[k], *rs[k], k + 1, score[k], r.a1a2, c.linelimit);
    f.write(j.outprefix + ".align" + std::to_string(k + 1));
  }
  pc.lap(CpuPhases::SMALL_FILES);
  // the report shows the decomposed basecalls
  BaseCalls bc = j.bc;
  bc.primary = j.primary;
  bc.secondary = j.secondary;
  bc.secDecompose = j.secdecomp;
  if (!r.bp.indelshift) r.bp.breakpoint = nearestSNP(j.trimLeft, j.trimRight, bc, findBestTraceSection(bc));  // centre on the first SNP
  ReportConfig rc;
  rc.trimLeft = (uint16_t)j.trimLeft;
  rc.trimRight = (uint16_t)j.trimRight;
  rc.qualCut = c.qualCut;
  rc.pratio = c.pratio;
  rc.genomeName = file_name(j.ref_path);
  rc.inputName = file_name(j.trace_path);
  if (c.callvariants) {
    TextBuf f(8192);
    std::vector<std::pair<std::string, uint64_t>> contigs;
    if (j.rs.filetype == 0) {
      const GenomeIndex* g = genome_index(c, j.ref_path);
      for (std::size_t i = 0; g && i < g->names.size(); ++i) contigs.emplace_back(g->names[i], (uint64_t)g->lengths[i] + 1);
    }
    vcfTextOutput(f, rc, bc, r.var, j.rs, j.rs.filetype == 0 ? &contigs : nullptr);
    f.write(j.outprefix + ".vcf");
    // ... and as the reference writes them: <prefix>.bcf (BGZF + BCF2.2 without htslib, bcf_out.hpp; no .csi index)
    if (!bcfOutput(j.outprefix + ".bcf", rc, bc, r.var, j.rs, j.rs.filetype == 0 ? &contigs : nullptr)) ++TextBuf::write_errors();  // (a deflate failure writes nothing: counted like a file that could not be written)
  }
  pc.lap(CpuPhases::PAD);
  TextBuf f(1 << 20);
  traceAlleleAlignJsonOut(f, rc, bc, j.tr, r);
  f.write(j.outprefix + ".json");
  pc.lap(CpuPhases::JSON);
}

int decompose_main(int argc, char** argv) {
  SageConfig c;
  if (parse(argc, argv, c)) {
    std::cout << "Usage: tracy " << argv[0] << " [OPTIONS] trace.ab1" << std::endl;
    usage_options(true);
    return -1;
  }
  if (c.trimStringency > 9) c.trimStringency = 9;
  if (!c.annotate.empty()) {
    std::cerr << "Variant annotation needs the Ensembl REST service and is not part of this build." << std::endl;
    return -1;
  }
  std::vector<Job> jobs;
  const bool batch = !c.batch.empty();
  if (int rc = collect_jobs(c, jobs)) return rc;
  for (Job const& j : jobs) {
    if (!regular_nonempty(j.trace_path)) {
      std::cerr << "Trace file is missing: " << j.trace_path << std::endl;
      return 1;
    }
    if (!regular_nonempty(j.ref_path)) {
      std::cerr << "Reference file is missing: " << file_name(j.ref_path) << std::endl;
      return 1;
    }
  }
  echo_command(argc, argv);
  std::cout << stamp() << "Load ab1 file" << std::endl;
  std::atomic<int> failed(0);
  const uint32_t nthreads = batch ? (c.threads ? c.threads : usable_cores()) : 1;
  StageTimes times;
  const uint32_t blk = std::max<uint32_t>(1u, batch ? block_size() : (uint32_t)jobs.size());
  std::vector<std::vector<std::unique_ptr<DecomposeGroup>>> block_groups((jobs.size() + blk - 1) / blk + 1);
  int fatal = 0;
  std::vector<int> rcs(jobs.size(), 0);
  auto prep = [&](uint32_t lo, uint32_t hi) {
    Stopwatch sw;
    for_each_index(hi - lo, nthreads, [&](uint32_t i) { rcs[lo + i] = prepare(c, jobs[lo + i], true); });
    for (uint32_t i = lo; i < hi; ++i) {
      if (rcs[i] != 0) {
        if (!batch) continue;
        std::cerr << "skipping " << jobs[i].trace_path << std::endl;
        ++failed;
      } else {
        jobs[i].ok = true;
      }
    }
    // the block's groups (reference kind, trims), packed for the device here -- host work that the device stage would otherwise do
    // between two calls, on the one thread every block has to pass
    std::map<std::pair<uint32_t, std::pair<uint32_t, uint32_t>>, std::vector<Job*>> groups;
    for (uint32_t i = lo; i < hi; ++i)
      if (jobs[i].ok) groups[std::make_pair((uint32_t)jobs[i].rs.filetype, std::make_pair(jobs[i].trimLeft, jobs[i].trimRight))].push_back(&jobs[i]);
    std::vector<std::unique_ptr<DecomposeGroup>>& mine = block_groups[lo / blk];
    for (auto& g : groups) {
      mine.emplace_back(new DecomposeGroup(std::move(g.second)));
      if (!mine.back()->wildtype) mine.back()->pack(c, nthreads);
    }
    times.add("read_basecall_profile_s", sw.seconds());
  };
  Device dev;
  bool dev_open = false;
  // the context is created (HIP start-up, ~0.1 s) while the first block is read
  std::future<int> dev_ready = std::async(std::launch::async, [&]() {
    Stopwatch sw;
    const int rc = dev.open(c.devices, 2);  // two chunks of a block in flight per GPU (small batches run on one lane)
    times.add("gpu_init_s", sw.seconds());
    return rc;
  });
  uint32_t blocks_done = 0;
  tracyhip_params prm{c.match, c.mismatch, c.gapopen, c.gapext, 1, 0};
  // (the progress lines of the reference are said once per command, with its first block)
  auto say = [&](const char* what) { if (blocks_done == 0) std::cout << stamp() << what << std::endl; };
  auto device = [&](uint32_t lo, uint32_t hi) -> bool {
    if (!batch && rcs[lo] != 0) { fatal = rcs[lo]; return false; }
    say("Find Reference Match");
    fatal = -1;
    if (!dev_open) {
      if (dev_ready.get() != 0) {
        gpu_fail("no usable GPU");
        return false;
      }
      dev_open = true;
    }
    Stopwatch sw;
    say("Alignment");
    for (auto& g : block_groups[lo / blk])
      if (!g->run(dev, c, prm, nthreads)) return false;
    block_groups[lo / blk].clear();  // (the packed batch and the raw results: everything the writers need is in the jobs now)
    say("InDel Search");
    std::vector<Job*> good;
    for (uint32_t i = lo; i < hi; ++i) {
      Job& j = jobs[i];
      if (!j.ok) continue;
      if (j.status != 0) {
        if (j.status == -1) std::cerr << "Alignment of trace to reference failed!" << std::endl;
        else if (j.status == -2) std::cerr << "No valid alignment found between consensus and reference!" << std::endl;
        else std::cerr << "Alignment too short between consensus and reference!" << std::endl;
        if (!batch) return false;
        std::cerr << "skipping " << j.trace_path << std::endl;
        j.ok = false;
        ++failed;
        continue;
      }
      good.push_back(&j);
    }
    say("Decompose Chromatogram");
    {  // (the reference's per-trace lines, said for the block at once: a flush per line is a system call per trace)
      std::ostringstream lines;
      for (Job* j : good) {
        if (j->dstatus.kind == 1)
          lines << "Complex mutation, decomposition: ins: " << j->dstatus.best_ins << ", del: " << j->dstatus.best_del
                << ", error: " << j->dstatus.best_fr << "\n";
        else if (j->dstatus.kind == 2)
          lines << "No InDel detected, traverse the whole alignment.\n";
      }
      std::cout << lines.str() << std::flush;
    }
    say("Estimate allelic fractions");
    say("Allele-specific alignments");
    if (c.callvariants) {
      say("Variant Calling");
      PhaseClock pv;
      if (!call_variants(dev.ctx, prm, good, nthreads)) return false;
      pv.lap(CpuPhases::VARIANTS);
    }
    fatal = 0;
    ++blocks_done;
    times.add("device_s", sw.seconds());
    return true;
  };
  auto write = [&](uint32_t lo, uint32_t hi) {
    Stopwatch sw;
    for_each_index(hi - lo, nthreads, [&](uint32_t i) {
      Job& j = jobs[lo + i];
      if (j.ok) write_decompose_outputs(c, j);
      if (batch) { Job done; done.trace_path.swap(j.trace_path); done.ok = j.ok; j = std::move(done); }  // the block's traces are released here
    });
    times.add("writers_s", sw.seconds());
  };
  if (!run_blocks((uint32_t)jobs.size(), blk, prep, device, write)) return fatal ? fatal : -1;
  times.report((uint32_t)jobs.size(), nthreads);
  std::cout << stamp() << "Done." << std::endl;
  const int rc_out = (failed || TextBuf::write_errors()) ? 2 : 0;  // (a file that could not be written counts like a trace that failed)
  if (batch) end_process(rc_out);
  return rc_out;
}

// ---- `tracy basecall` (teal.h:24-117): host only, no device needed -------------------------------------------
int basecall_main(int argc, char** argv) {
  float pratio = 0.33f, trimStringency = 0;
  uint16_t trimLeft = 0, trimRight = 0;
  std::string format = "json", otype = "primary", outfile = "out.json", tracein;
  static const std::map<std::string, char> longs = {{"help", '?'}, {"pratio", 'p'}, {"format", 'f'}, {"otype", 'y'}, {"trim", 't'},
                                                    {"trimLeft", 'q'}, {"trimRight", 'u'}, {"output", 'o'}};
  bool bad = false;
  for (int i = 1; i < argc && !bad; ++i) {
    std::string a = argv[i], val;
    char opt = 0;
    bool has_val = false;
    if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
      std::string name = a.substr(2);
      const std::size_t eq = name.find('=');
      if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
      auto it = longs.find(name);
      if (it == longs.end()) { std::cerr << "unrecognised option '" << a << "'" << std::endl; bad = true; break; }
      opt = it->second;
    } else if (a.size() >= 2 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9')) {
      opt = a[1];
      if (a.size() > 2) { val = a.substr(2); has_val = true; }
    } else {
      tracein = a;
      continue;
    }
    if (opt == '?') { bad = true; break; }
    if (!has_val) {
      if (i + 1 >= argc) { std::cerr << "the required argument for option '" << a << "' is missing" << std::endl; bad = true; break; }
      val = argv[++i];
    }
    switch (opt) {
      case 'p': pratio = std::strtof(val.c_str(), nullptr); break;
      case 'f': format = val; break;
      case 'y': otype = val; break;
      case 't': trimStringency = std::strtof(val.c_str(), nullptr); break;
      case 'q': trimLeft = (uint16_t)std::atoi(val.c_str()); break;
      case 'u': trimRight = (uint16_t)std::atoi(val.c_str()); break;
      case 'o': outfile = val; break;
      default: std::cerr << "unrecognised option '" << a << "'" << std::endl; bad = true; break;
    }
  }
  if (bad || tracein.empty()) {
    std::cout << "Usage: tracy " << argv[0] << " [OPTIONS] trace.ab1" << std::endl;
    std::cout << "Generic options:\n"
                 "  -? [ --help ]                    show help message\n"
                 "  -p [ --pratio ] arg (=0.33)      peak ratio to call a base\n"
                 "  -f [ --format ] arg (=json)      output format [json|tsv|fasta|fastq]\n"
                 "  -y [ --otype ] arg (=primary)    fasta/fastq sequence [primary|secondary|consensus]\n"
                 "  -t [ --trim ] arg (=0)           trimming stringency [1:9], 0: use trimLeft and trimRight\n"
                 "  -q [ --trimLeft ] arg (=0)       trim size left\n"
                 "  -u [ --trimRight ] arg (=0)      trim size right\n"
                 "  -o [ --output ] arg (=out.json)  basecalling output\n\n";
    return -1;
  }
  echo_command(argc, argv);
  if (!regular_nonempty(tracein)) {
    std::cerr << "Input trace file is missing: " << tracein << std::endl;
    return 1;
  }
  Trace tr;
  if (!load_trace(tracein, tr)) return -1;
  BaseCalls bc;
  basecall(tr, bc, pratio);
  if (trimStringency >= 1) {
    uint32_t l = 0, r = 0;
    trimTrace(trimStringency, bc, l, r);
    trimLeft = (uint16_t)l;
    trimRight = (uint16_t)r;
  }
  if ((uint32_t)trimLeft + trimRight >= bc.bcPos.size()) {
    std::cerr << "The sum of the left and right trim size is larger than the trace!" << std::endl;
    return -1;
  }
  if (format == "tsv") {
    traceTxtOut(std::string(outfile), bc, tr, trimLeft, trimRight);
  } else if (format == "fasta" || format == "fastq") {  // traceFastaOut / traceFastqOut, fasta.h:98-155
    std::ofstream f(outfile.c_str());
    std::string const* seq = otype == "primary" ? &bc.primary : otype == "secondary" ? &bc.secondary : otype == "consensus" ? &bc.consensus : nullptr;
    if (seq) {
      f << (format == "fasta" ? ">" : "@") << otype << std::endl;
      for (uint32_t i = trimLeft; i < seq->size() - trimRight; ++i) f << (*seq)[i];
      f << std::endl;
    }
    if (format == "fastq") {
      f << "+" << std::endl;
      uint32_t call = 0;
      int32_t next = bc.bcPos[0];
      for (int32_t x = 0; x < (int32_t)tr.traceACGT[0].size(); ++x) {
        if (next != x) continue;
        if (call >= trimLeft && call < bc.primary.size() - trimRight) f << (char)(bc.estQual[call] + 33);
        if (call < bc.bcPos.size() - 1) next = bc.bcPos[++call];
      }
      f << std::endl;
    }
  } else {  // traceJsonOut, json.h:96-105
    std::ofstream f(outfile.c_str());
    f << "{" << std::endl;
    traceJsonBody(f, bc, tr);
    f << std::endl << "}" << std::endl;
  }
  std::cout << stamp() << "Done." << std::endl;
  return 0;
}

// ---- `tracy index` (index.h:36-124): the genome's k-mer table on disk ---------------------------------------------------
int index_main(int argc, char** argv) {
  std::string outfile, genome;
  uint32_t kmer = 15;
  bool bad = false;
  for (int i = 1; i < argc && !bad; ++i) {
    const std::string a = argv[i];
    auto value = [&](std::string& dst) { if (i + 1 < argc) dst = argv[++i]; else bad = true; };
    std::string v;
    if (a == "-o" || a == "--output") value(outfile);
    else if (a == "-k" || a == "--kmer") { value(v); kmer = (uint32_t)std::atoi(v.c_str()); }
    else if (a == "-?" || a == "--help") bad = true;
    else if (!a.empty() && a[0] == '-') bad = true;
    else genome = a;
  }
  if (bad || genome.empty() || kmer < 1 || kmer > 32) {
    std::cout << "Usage: tracy index [OPTIONS] genome.fa.gz" << std::endl;
    std::cout << "  -o [ --output ] arg      output file (default: <genome>.tidx, found by align / decompose -r <genome>)" << std::endl;
    std::cout << "  -k [ --kmer ] arg (=15)  k-mer size of the table (align / decompose -k must match)" << std::endl;
    return -1;
  }
  if (!regular_nonempty(genome)) {
    std::cerr << "Input reference file is missing: " << genome << std::endl;
    return 1;
  }
  if (outfile.empty()) outfile = genome + ".tidx";
  echo_command(argc, argv);
  std::cout << stamp() << "Load genome" << std::endl;
  GenomeIndex g;
  if (!g.load(genome)) {
    std::cerr << "Couldn't recognize reference file format!" << std::endl;
    return 1;
  }
  std::cout << stamp() << "Create index" << std::endl;
  g.build(kmer);
  if (!g.save(outfile)) {
    std::cerr << "Cannot write " << outfile << std::endl;
    return 1;
  }
  std::cout << stamp() << "Done." << std::endl;
  return 0;
}

#include "assemble_cli.inc"
#include "consensus_cli.inc"

}  // namespace

int main(int argc, char** argv) {
  tracyhip_tune_host_allocator();  // (process-wide allocator settings are the application's to choose: this process is one)
  if (argc >= 2 && std::strcmp(argv[1], "align") == 0) return align_main(argc - 1, argv + 1);
  if (argc >= 2 && std::strcmp(argv[1], "decompose") == 0) return decompose_main(argc - 1, argv + 1);
  if (argc >= 2 && std::strcmp(argv[1], "assemble") == 0) return assemble_main(argc - 1, argv + 1);
  if (argc >= 2 && std::strcmp(argv[1], "basecall") == 0) return basecall_main(argc - 1, argv + 1);
  if (argc >= 2 && std::strcmp(argv[1], "consensus") == 0) return consensus_main(argc - 1, argv + 1);
  if (argc >= 2 && std::strcmp(argv[1], "index") == 0) return index_main(argc - 1, argv + 1);
  std::cout << "Usage: tracy_amd_cli align|decompose [OPTIONS] -r genome.fa trace.ab1" << std::endl;
  std::cout << "       tracy_amd_cli align|decompose [OPTIONS] --batch manifest.tsv" << std::endl;
  std::cout << "       tracy_amd_cli assemble [OPTIONS] [-r reference.fa] trace1.ab1 trace2.ab1 ..." << std::endl;
  std::cout << "       tracy_amd_cli basecall [OPTIONS] trace.ab1" << std::endl;
  std::cout << "       tracy_amd_cli consensus [OPTIONS] trace1.ab1 trace2.ab1" << std::endl;
  std::cout << "       tracy_amd_cli index [-o genome.tidx] [-k 15] genome.fa.gz" << std::endl;
  return argc < 2 ? 0 : 1;
}
