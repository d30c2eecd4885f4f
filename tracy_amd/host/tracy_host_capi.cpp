// tracy_host_capi.cpp -- C wrappers over tracy_host.hpp + the seeded synthetic workload generator used
// by bench.py and the parity tests.  Host-only library (libtracy_host.so):
// basecalling / profile creation are host stages in the reference too.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "bcf_out.hpp"
#include "indigo_out.hpp"
#include "sage_out.hpp"
#include "seed.hpp"
#include "trace_io.hpp"
#include "tracy_host.hpp"

using namespace tracy_amd;

namespace {

struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return n ? (uint32_t)(next() % n) : 0; }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

const char kBases[4] = {'A', 'C', 'G', 'T'};
inline int base_index(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }
inline char complement(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }

// add a triangular peak (half-width 5 samples) of height amp centred at x
void add_peak(std::vector<int32_t>& ch, int32_t x, double amp) {
  for (int d = -5; d <= 5; ++d) {
    const int32_t i = x + d;
    if (i < 0 || i >= (int32_t)ch.size()) continue;
    ch[i] += (int32_t)(amp * (1.0 - std::abs(d) / 6.0));
  }
}

// One `tracy align` case: a uniform ACGT window of n bases, a trace of mf bases copied from it at a
// random offset (forward or reverse-complement, 1 % substitutions, 0.2 % 1-3 bp indels), rendered as a
// 4-channel chromatogram with 12 samples per base, primary amplitude U[400,1200] and a background peak
// of 5-15 % in another channel.  hetero > 0 adds a second allele (see synth_decompose_case).
void synth_sequence(SplitMix64& rng, uint32_t n, uint32_t mf, std::string& ref, std::string& seq, bool& reverse) {
  ref.resize(n);
  for (uint32_t i = 0; i < n; ++i) ref[i] = kBases[rng.below(4)];
  const uint32_t span = (n > mf + 40) ? n - mf - 40 : 0;
  uint32_t lo = std::min<uint32_t>(500, span / 2);
  const uint32_t start = lo + rng.below(span - 2 * lo + 1);
  reverse = (rng.next() & 1) != 0;
  std::string src = ref.substr(start, std::min<uint32_t>(n - start, mf + 40));
  if (reverse) {
    std::reverse(src.begin(), src.end());
    for (auto& c : src) c = complement(c);
  }
  seq.clear();
  for (size_t i = 0; i < src.size() && seq.size() < mf; ++i) {
    const double u = rng.unit();
    if (u < 0.001) { i += rng.below(3); continue; }                                                  // deletion
    if (u < 0.002) { const uint32_t k = 1 + rng.below(3); for (uint32_t j = 0; j < k; ++j) seq.push_back(kBases[rng.below(4)]); }  // insertion
    if (u > 0.99) seq.push_back(kBases[rng.below(4)]);                                               // substitution
    else seq.push_back(src[i]);
  }
  while (seq.size() < mf) seq.push_back(kBases[rng.below(4)]);
  seq.resize(mf);
}

void render_trace(SplitMix64& rng, std::string const& allele1, std::string const* allele2, double frac1, Trace& tr) {
  const size_t len = allele2 ? std::max(allele1.size(), allele2->size()) : allele1.size();
  const size_t samples = 12 * len + 12;
  tr.traceACGT.assign(4, std::vector<int32_t>(samples, 0));
  tr.basecallpos.resize(len);
  for (size_t j = 0; j < len; ++j) {
    const int32_t x = (int32_t)(6 + 12 * j);
    tr.basecallpos[j] = x;
    const double amp = 400.0 + 800.0 * rng.unit();
    const double bg = amp * (0.05 + 0.10 * rng.unit());
    int used[2] = {-1, -1};
    if (j < allele1.size()) { used[0] = base_index(allele1[j]); add_peak(tr.traceACGT[used[0]], x, allele2 ? amp * frac1 : amp); }
    if (allele2 && j < allele2->size()) { used[1] = base_index((*allele2)[j]); add_peak(tr.traceACGT[used[1]], x, amp * (1.0 - frac1)); }
    int b = (int)rng.below(4);
    while (b == used[0] || b == used[1]) b = (b + 1) & 3;
    add_peak(tr.traceACGT[b], x, bg);
  }
}

}  // namespace

extern "C" {

// flat-array wrappers -----------------------------------------------------------------------------------
size_t tracyhost_basecall(const int32_t* trace, size_t nsamples, const int32_t* basecallpos, size_t npos, float sigratio,
                          char* primary, char* secondary, char* consensus, int32_t* bcpos) {
  Trace tr;
  tr.traceACGT.resize(4);
  for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(trace + k * nsamples, trace + (k + 1) * nsamples);
  tr.basecallpos.assign(basecallpos, basecallpos + npos);
  BaseCalls bc;
  basecall(tr, bc, sigratio);
  const size_t n = bc.primary.size();
  std::memcpy(primary, bc.primary.data(), n);
  std::memcpy(secondary, bc.secondary.data(), n);
  std::memcpy(consensus, bc.consensus.data(), n);
  std::memcpy(bcpos, bc.bcPos.data(), n * sizeof(int32_t));
  return n;
}

int32_t tracyhost_create_profile(const int32_t* trace, size_t nsamples, const int32_t* bcpos, const char* primary,
                                 const char* secondary, size_t nbc, int32_t trimleft, int32_t trimright, float* out) {
  Trace tr;
  tr.traceACGT.resize(4);
  for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(trace + k * nsamples, trace + (k + 1) * nsamples);
  BaseCalls bc;
  bc.bcPos.assign(bcpos, bcpos + nbc);
  bc.primary.assign(primary, nbc);
  bc.secondary.assign(secondary, nbc);
  Profile p;
  createProfile(tr, bc, p, trimleft, trimright);
  std::memcpy(out, p.data(), sizeof(float) * 6 * p.cols);
  return (int32_t)p.cols;
}

char tracyhost_iupac(char a, char b) { return iupac(a, b); }

// basecall + the per-base quality estimate it ends with (abif.h:232-253, 510)
size_t tracyhost_basecall_qual(const int32_t* trace, size_t nsamples, const int32_t* basecallpos, size_t npos, float sigratio,
                               char* primary, char* secondary, char* consensus, int32_t* bcpos, uint8_t* estqual) {
  Trace tr;
  tr.traceACGT.resize(4);
  for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(trace + k * nsamples, trace + (k + 1) * nsamples);
  tr.basecallpos.assign(basecallpos, basecallpos + npos);
  BaseCalls bc;
  basecall(tr, bc, sigratio);
  const size_t n = bc.primary.size();
  std::memcpy(primary, bc.primary.data(), n);
  std::memcpy(secondary, bc.secondary.data(), n);
  std::memcpy(consensus, bc.consensus.data(), n);
  std::memcpy(bcpos, bc.bcPos.data(), n * sizeof(int32_t));
  std::memcpy(estqual, bc.estQual.data(), n);
  return n;
}

// chromatogram files: format 0 = ABIF, 1 = SCF (scf.h:18-34).  The handle owns a Trace.
void* tracyhost_trace_read(const char* path, int32_t* format) {
  const int32_t ft = traceFormat(path);
  if (format) *format = ft;
  Trace* tr = new Trace();
  const bool ok = ft == 0 ? readab(path, *tr) : ft == 1 ? readscf(path, *tr) : false;
  if (!ok) { delete tr; return nullptr; }
  return tr;
}
void tracyhost_trace_dims(const void* h, uint64_t* nsamples, uint64_t* ncalls) {
  const Trace* tr = static_cast<const Trace*>(h);
  size_t ns = 0;
  for (auto const& c : tr->traceACGT) ns = std::max(ns, c.size());
  *nsamples = ns;
  *ncalls = tr->basecallpos.size();
}
// signal: [4][nsamples] (channels shorter than nsamples are zero padded); text outputs have ncalls bytes
void tracyhost_trace_get(const void* h, int32_t* signal, int32_t* basecallpos, char* basecalls1, char* basecalls2, uint8_t* qual) {
  const Trace* tr = static_cast<const Trace*>(h);
  uint64_t ns, nc;
  tracyhost_trace_dims(h, &ns, &nc);
  for (size_t k = 0; k < 4; ++k)
    for (size_t i = 0; i < ns; ++i) signal[k * ns + i] = (k < tr->traceACGT.size() && i < tr->traceACGT[k].size()) ? tr->traceACGT[k][i] : 0;
  std::memcpy(basecallpos, tr->basecallpos.data(), nc * sizeof(int32_t));
  if (basecalls1) std::memcpy(basecalls1, tr->basecalls1.data(), std::min<size_t>(nc, tr->basecalls1.size()));
  if (basecalls2) std::memcpy(basecalls2, tr->basecalls2.data(), std::min<size_t>(nc, tr->basecalls2.size()));
  if (qual) std::memcpy(qual, tr->qual.data(), std::min<size_t>(nc, tr->qual.size()));
}
void tracyhost_trace_free(void* h) { delete static_cast<Trace*>(h); }

// the build's ABIF writer: signal [4][nsamples] in A,C,G,T order, written in dye order `order`
int32_t tracyhost_writeab(const char* path, const int32_t* signal, size_t nsamples, const int32_t* peaks, size_t npeaks,
                          const char* primary, size_t nprimary, const uint8_t* qual, size_t nqual, const char* secondary,
                          size_t nsecondary, const char* order) {
  Trace::TACGTMountains acgt(4);
  for (int k = 0; k < 4; ++k) acgt[k].assign(signal + k * nsamples, signal + (k + 1) * nsamples);
  return writeab(path, acgt, std::vector<int32_t>(peaks, peaks + npeaks), std::string(primary, nprimary),
                 std::vector<uint8_t>(qual, qual + nqual), secondary ? std::string(secondary, nsecondary) : std::string(),
                 order ? order : "GATC") ? 0 : -1;
}

// basecall + traceTxtOut (abif.h:513-533) into `outfile`
int32_t tracyhost_trace_txt(const char* outfile, const int32_t* trace, size_t nsamples, const int32_t* basecallpos, size_t npos,
                            float sigratio, uint32_t left_trim, uint32_t right_trim) {
  Trace tr;
  tr.traceACGT.resize(4);
  for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(trace + k * nsamples, trace + (k + 1) * nsamples);
  tr.basecallpos.assign(basecallpos, basecallpos + npos);
  BaseCalls bc;
  basecall(tr, bc, sigratio);
  if (bc.bcPos.empty()) return -1;
  traceTxtOut(std::string(outfile), bc, tr, left_trim, right_trim);
  return 0;
}

// the three alignment files of `tracy align` (sage.h:313-345) for a trace given as arrays and its final
// alignment rows: <prefix>.align.fa, <prefix>.txt, <prefix>.json
int32_t tracyhost_align_outputs(const char* prefix, const char* trace_stem, const int32_t* trace, size_t nsamples,
                                const int32_t* basecallpos, size_t npos, float sigratio, const char* row0, const char* row1,
                                size_t cols, const char* chr, const char* refslice, size_t nref, uint32_t pos, int32_t forward,
                                int32_t score, uint32_t linelimit) {
  Trace tr;
  tr.traceACGT.resize(4);
  for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(trace + k * nsamples, trace + (k + 1) * nsamples);
  tr.basecallpos.assign(basecallpos, basecallpos + npos);
  BaseCalls bc;
  basecall(tr, bc, sigratio);
  if (bc.bcPos.empty()) return -1;
  AlignRows rows;
  rows.row0.assign(row0, cols);
  rows.row1.assign(row1, cols);
  ReferenceSlice rs;
  rs.forward = forward != 0;
  rs.pos = pos;
  rs.chr = chr;
  rs.refslice.assign(refslice, nref);
  PaddedTrace padded;
  alignmentTracePadding(rows.row0, tr, bc, padded);
  const std::string pre(prefix);
  {
    std::ofstream f((pre + ".align.fa").c_str());
    alignFastaOut(f, trace_stem, rs, rows);
  }
  plotAlignment(pre + ".txt", rows, rs, score, linelimit);
  traceAlignJsonOut(pre + ".json", padded, rs, rows);
  return 0;
}

// the files `tracy decompose` writes after the device chain (indigo.h:340-442): <prefix>.decomp, .align1,
// .align2, .align3, .json and, with call_variants, <prefix>.vcf.  The trace is given as arrays (basecalled
// here, then overlaid with the decomposed calls); alignments as gapped rows.
struct tracyhost_decompose_report {
  const char* prefix;
  const char* genome_name;
  const char* input_name;
  const int32_t* trace;
  uint64_t nsamples;
  const int32_t* basecallpos;
  uint64_t npos;
  float pratio;
  uint32_t trim_left, trim_right, qual_cut, linelimit;
  const char* primary;    // decomposed calls, ncalls bytes each
  const char* secondary;
  const char* secdecomp;
  uint64_t ncalls;
  const char* rows[3][2]; // final1, final2, final3
  uint64_t cols[3];
  const char* var_rows[2][2];  // alignments variants are called on (the reverse-complement ones for reverse traces)
  uint64_t var_cols[2];
  int32_t call_variants;
  const char* chr;
  int32_t forward;
  uint32_t pos[2];        // allele1.pos, allele2.pos
  uint64_t slice_len[2];  // allele1/2 .refslice.size()
  uint64_t ref_len;       // rs.refslice.size()
  int32_t score[3];
  int32_t indelshift;
  uint32_t breakpoint;
  double a1, a2;
  const int32_t* dcp_indel;
  const int32_t* dcp_err;
  uint64_t dcp_n;
};

int32_t tracyhost_decompose_outputs(const tracyhost_decompose_report* rp) {
  Trace tr;
  tr.traceACGT.resize(4);
  for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(rp->trace + k * rp->nsamples, rp->trace + (k + 1) * rp->nsamples);
  tr.basecallpos.assign(rp->basecallpos, rp->basecallpos + rp->npos);
  BaseCalls bc;
  basecall(tr, bc, rp->pratio);
  if (bc.bcPos.size() != rp->ncalls) return -1;
  bc.primary.assign(rp->primary, rp->ncalls);
  bc.secondary.assign(rp->secondary, rp->ncalls);
  bc.secDecompose.assign(rp->secdecomp, rp->ncalls);
  AlleleReport r;
  AlignRows* al[3] = {&r.align1, &r.align2, &r.align3};
  for (int k = 0; k < 3; ++k) {
    al[k]->row0.assign(rp->rows[k][0], rp->cols[k]);
    al[k]->row1.assign(rp->rows[k][1], rp->cols[k]);
  }
  ReferenceSlice rs;
  rs.chr = rp->chr;
  rs.forward = rp->forward != 0;
  rs.filetype = 1;
  rs.refslice.assign(rp->ref_len, 'N');
  ReferenceSlice* slot[2] = {&r.rs1, &r.rs2};
  for (int k = 0; k < 2; ++k) {
    *slot[k] = rs;
    slot[k]->pos = rp->pos[k];
    slot[k]->refslice.assign(rp->slice_len[k], 'N');
  }
  r.a1Score = rp->score[0]; r.a2Score = rp->score[1]; r.a3Score = rp->score[2];
  r.bp.indelshift = rp->indelshift != 0;
  r.bp.breakpoint = rp->breakpoint;
  r.a1a2 = std::make_pair(rp->a1, rp->a2);
  for (uint64_t i = 0; i < rp->dcp_n; ++i) r.dcp.emplace_back(rp->dcp_indel[i], rp->dcp_err[i]);
  const std::string pre(rp->prefix);
  {
    std::ofstream f((pre + ".decomp").c_str());
    writeDecomposition(f, r.dcp);
  }
  ReferenceSlice secrs;
  secrs.refslice = trimmedSeq(bc.secDecompose, rp->trim_left, rp->trim_right);
  secrs.forward = true;
  secrs.chr = "Alt2";
  ReferenceSlice const* prs[3] = {&r.rs1, &r.rs2, &secrs};
  for (int k = 0; k < 3; ++k) {
    std::ofstream f((pre + ".align" + std::to_string(k + 1)).c_str());
    plotAlignment(f, *al[k], *prs[k], k + 1, rp->score[k], r.a1a2, rp->linelimit);
  }
  if (!r.bp.indelshift) r.bp.breakpoint = nearestSNP(rp->trim_left, rp->trim_right, bc, findBestTraceSection(bc));
  ReportConfig rc;
  rc.trimLeft = (uint16_t)rp->trim_left;
  rc.trimRight = (uint16_t)rp->trim_right;
  rc.qualCut = (uint16_t)rp->qual_cut;
  rc.pratio = rp->pratio;
  rc.genomeName = rp->genome_name;
  rc.inputName = rp->input_name;
  if (rp->call_variants) {
    for (int k = 0; k < 2; ++k) {
      AlignRows v;
      v.row0.assign(rp->var_rows[k][0], rp->var_cols[k]);
      v.row1.assign(rp->var_rows[k][1], rp->var_cols[k]);
      ReferenceSlice vrs = *slot[k];
      if (!rs.forward) reverseReferenceSlice(*slot[k], vrs);
      callVariants(v, vrs, r.var);
    }
    std::sort(r.var.begin(), r.var.end());
    std::ofstream f((pre + ".vcf").c_str());
    vcfTextOutput(f, rc, bc, r.var, rs);
    if (!bcfOutput(pre + ".bcf", rc, bc, r.var, rs)) return -2;
  }
  std::ofstream f((pre + ".json").c_str());
  traceAlleleAlignJsonOut(f, rc, bc, tr, r);
  return 0;
}

// ---- k-mer seeding in an indexed genome (seed.hpp; fmindex.h:173-326) --------------------------------
// path: a (gzip-compressed) multi-FASTA -- the table is built in memory -- or an index file written by tracyhost_genome_save
// (`tracy_amd_cli index`), which is mapped read-only (kmer must be the index's)
void* tracyhost_genome_open(const char* path, uint32_t kmer, uint32_t nthreads) {
  GenomeIndex* g = new GenomeIndex();
  if (GenomeIndex::is_index_file(path)) {
    if (!g->open_index(path) || g->k != kmer) { delete g; return nullptr; }
    return g;
  }
  if (!g->load(path)) { delete g; return nullptr; }
  g->build(kmer, nthreads);
  return g;
}
int tracyhost_genome_save(const void* h, const char* path) { return static_cast<const GenomeIndex*>(h)->save(path) ? 0 : -1; }
void tracyhost_genome_free(void* h) { delete static_cast<GenomeIndex*>(h); }
uint32_t tracyhost_genome_contigs(const void* h) { return (uint32_t)static_cast<const GenomeIndex*>(h)->names.size(); }
uint64_t tracyhost_genome_count(const void* h, const char* pat, size_t n) { return static_cast<const GenomeIndex*>(h)->count(std::string(pat, n)); }

// getReferenceSlice for a batch of consensus strings (trace t: consensus + cons_off[t], cons_len[t] bytes), one
// thread per stripe of traces.  Per trace: status 1 = anchored, 0 = not; forward, kmersupport, pos (window start
// in the contig), contig index, and the ORIENTED window at slices + t * slice_cap (slice_len[t] bytes).
void tracyhost_seed_batch(const void* h, uint32_t ntraces, const char* consensus, const uint64_t* cons_off, const uint32_t* cons_len,
                          uint32_t trim_left, uint32_t trim_right, uint32_t kmer, uint32_t min_support, uint32_t maxindel,
                          uint32_t nthreads, int32_t* status, uint8_t* forward, uint32_t* kmersupport, uint32_t* pos, uint32_t* contig,
                          char* slices, uint64_t slice_cap, uint32_t* slice_len) {
  const GenomeIndex* g = static_cast<const GenomeIndex*>(h);
  // seeding is a stream of dependent table look-ups (memory latency, not arithmetic): every hardware thread helps,
  // also under a CPU quota (measured on the 256-thread / 16-CPU GPU box: 256 threads 171 k traces/s, 16 threads 99 k)
  if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
  SeedConfig sc;
  sc.trimLeft = (uint16_t)trim_left; sc.trimRight = (uint16_t)trim_right; sc.kmer = (uint16_t)kmer;
  sc.minKmerSupport = (uint16_t)min_support; sc.maxindel = (uint16_t)maxindel;
  auto work = [&](uint32_t tid) {
    for (uint32_t t = tid; t < ntraces; t += nthreads) {
      ReferenceSlice rs;
      rs.filetype = 0;
      const bool ok = getReferenceSlice(sc, *g, std::string(consensus + cons_off[t], cons_len[t]), rs);
      status[t] = ok ? 1 : 0;
      slice_len[t] = 0;
      if (!ok) continue;
      forward[t] = rs.forward ? 1 : 0;
      kmersupport[t] = rs.kmersupport;
      pos[t] = rs.pos;
      uint32_t ci = 0;
      while (ci < g->names.size() && g->names[ci] != rs.chr) ++ci;
      contig[t] = ci;
      const size_t n = std::min<size_t>(rs.refslice.size(), slice_cap);
      std::memcpy(slices + (size_t)t * slice_cap, rs.refslice.data(), n);
      slice_len[t] = (uint32_t)n;
    }
  };
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < nthreads; ++t) th.emplace_back(work, t);
  for (auto& t : th) t.join();
}

// trimTrace (trim.h:35-73) of the basecalled trace
int32_t tracyhost_trim_trace(const int32_t* trace, size_t nsamples, const int32_t* basecallpos, size_t npos, float sigratio,
                             float stringency, uint32_t* left, uint32_t* right) {
  Trace tr;
  tr.traceACGT.resize(4);
  for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(trace + k * nsamples, trace + (k + 1) * nsamples);
  tr.basecallpos.assign(basecallpos, basecallpos + npos);
  BaseCalls bc;
  basecall(tr, bc, sigratio);
  trimTrace(stringency, bc, *left, *right);
  return 0;
}

// loadSingleFasta (fasta.h:54-95): returns the sequence length, -1 on error; name/seq need capacity
int64_t tracyhost_load_fasta(const char* path, char* name, size_t name_cap, char* seq, size_t seq_cap) {
  std::string n, s;
  if (!loadSingleFasta(path, n, s)) return -1;
  if (n.size() + 1 > name_cap || s.size() > seq_cap) return -2;
  std::memcpy(name, n.c_str(), n.size() + 1);
  std::memcpy(seq, s.data(), s.size());
  return (int64_t)s.size();
}

// Seeded synthetic `tracy align` workload: ntraces cases, case i seeded with seed0 + i.  refs: ntraces x n
// bytes; profiles: ntraces x 6 x mf floats (createProfile of the basecalled synthetic chromatogram; when
// the basecaller drops a window the profile is padded with uniform 0.25 columns to keep mf columns).
// reverse[i] = 1 when the trace was copied from the reverse strand.  Multi-threaded over cases.
void tracyhost_synth_align(uint64_t seed0, uint32_t ntraces, uint32_t n, uint32_t mf, uint8_t* refs, float* profiles,
                           uint8_t* reverse, uint32_t nthreads) {
  if (nthreads == 0) nthreads = tracy_amd::usable_threads();
  auto work = [&](uint32_t tid) {
    for (uint32_t i = tid; i < ntraces; i += nthreads) {
      SplitMix64 rng(seed0 + i);
      std::string ref, seq;
      bool rev;
      synth_sequence(rng, n, mf, ref, seq, rev);
      Trace tr;
      render_trace(rng, seq, nullptr, 1.0, tr);
      BaseCalls bc;
      basecall(tr, bc, 0.33f);
      Profile p;
      createProfile(tr, bc, p, 0, 0);
      std::memcpy(refs + (size_t)i * n, ref.data(), n);
      float* dst = profiles + (size_t)i * 6 * mf;
      for (int k = 0; k < 6; ++k)
        for (uint32_t j = 0; j < mf; ++j) dst[(size_t)k * mf + j] = (j < p.cols) ? p(k, j) : (k < 4 ? 0.25f : 0.0f);
      if (reverse) reverse[i] = rev ? 1 : 0;
    }
  };
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < nthreads; ++t) th.emplace_back(work, t);
  for (auto& t : th) t.join();
}


// Seeded synthetic `tracy decompose` case: reference window of n bases; allele 1 =
// mf bases copied from it (forward strand), allele 2 = allele 1 with one heterozygous indel (length
// U[1,maxlen], insertion or deletion) at a position U[150, mf-300] plus 0.5 % heterozygous SNVs; the two
// alleles are mixed frac1 : 1-frac1 in the chromatogram.  kind & 3: 0 = het indel, 1 = het SNVs only (no indel), 2 = homozygous
// indel only (both alleles carry it, no het SNVs), 3 = no variant at all.  kind & 32: a low-complexity window.  kind & 16: the window handed out is the reverse
// complement of the one the trace was copied from (the trace reads the reverse strand of its reference).
// Outputs: ref[n]; signal[4][12*(mf+40)+12] (zero padded), basecallpos[npos]; returns npos.
uint32_t tracyhost_synth_decompose(uint64_t seed, uint32_t n, uint32_t mf, uint32_t maxlen, int kind, double frac1,
                                   uint8_t* ref_out, int32_t* signal, uint32_t nsamples_cap, int32_t* basecallpos,
                                   int32_t* indel_out) {
  SplitMix64 rng(seed);
  std::string ref(n, 'A');
  if (kind & 32) {  // low complexity: short units repeated a few times -- alignments with many co-optimal paths (gaps that can sit anywhere in a run)
    for (uint32_t i = 0; i < n;) {
      char unit[6];
      const uint32_t ul = 1 + rng.below(6), reps = 1 + rng.below(ul == 1 ? 8 : 5);
      for (uint32_t u = 0; u < ul; ++u) unit[u] = kBases[rng.below(4)];
      for (uint32_t r = 0; r < reps && i < n; ++r)
        for (uint32_t u = 0; u < ul && i < n; ++u) ref[i++] = unit[u];
    }
  } else {
    for (uint32_t i = 0; i < n; ++i) ref[i] = kBases[rng.below(4)];
  }
  const uint32_t start = (n > mf + 200) ? 100 + rng.below(n - mf - 200) : 0;
  std::string a1 = ref.substr(start, mf + 40);
  std::string a2 = a1;
  int32_t indel = 0;
  const bool reverse_strand = (kind & 16) != 0;
  kind &= 3;
  if (kind == 0 || kind == 2) {
    const uint32_t pos = 150 + rng.below(mf > 450 ? mf - 450 : 1);
    const uint32_t len = 1 + rng.below(maxlen);
    if (rng.next() & 1) {  // deletion in allele 2
      a2.erase(pos, len);
      indel = -(int32_t)len;
    } else {
      std::string ins(len, 'A');
      for (auto& c : ins) c = kBases[rng.below(4)];
      a2.insert(pos, ins);
      indel = (int32_t)len;
    }
    if (kind == 2) a1 = a2;  // homozygous: both alleles carry the indel
  }
  if (kind < 2)
    for (size_t i = 0; i < a2.size(); ++i)
      if (rng.unit() < 0.005) a2[i] = kBases[(base_index(a2[i]) + 1 + rng.below(3)) & 3];
  if (reverse_strand) {
    std::reverse(ref.begin(), ref.end());
    for (auto& c : ref) c = complement(c);
  }
  a1.resize(mf);
  a2.resize(mf);
  Trace tr;
  render_trace(rng, a1, &a2, frac1, tr);
  const size_t ns = tr.traceACGT[0].size();
  for (int k = 0; k < 4; ++k)
    for (uint32_t i = 0; i < nsamples_cap; ++i) signal[(size_t)k * nsamples_cap + i] = i < ns ? tr.traceACGT[k][i] : 0;
  for (size_t i = 0; i < tr.basecallpos.size(); ++i) basecallpos[i] = tr.basecallpos[i];
  std::memcpy(ref_out, ref.data(), n);
  if (indel_out) *indel_out = indel;
  return (uint32_t)tr.basecallpos.size();
}


// Batch of synthetic `tracy decompose` inputs (config 3), multi-threaded: trace i is seeded with seed0 + i,
// 80 % heterozygous indels (kind 0), 20 % SNV-only (kind 1), allele mix 60/40.  Every trace is basecalled
// (0.33) and profiled here; traces whose basecaller dropped a window are regenerated with the next seed
// stride so that all traces have exactly mf basecalls.  Layouts: refs [nt][n]; signal [nt][4][ns] with
// ns = 12*mf+12; bcpos / primary / secondary [nt][mf]; profiles [nt][6][mf].
// mix 0: the round-1 batch above (all forward strand).  mix 1: BASELINE configs[2] as SURVEY.md 8d words it -- trace i with
// i % 10 == 8 carries a homozygous indel only, i % 10 == 9 no variant, the rest one het indel + 0.5 % het SNVs; odd traces
// read the reverse strand of their window.  mix 2: mix 1 in low-complexity windows (short tandem repeats: co-optimal alignments everywhere).
static void synth_decompose_batch_mix(uint64_t seed0, uint32_t nt, uint32_t n, uint32_t mf, uint8_t* refs, int32_t* signal,
                                      int32_t* bcpos, uint8_t* primary, uint8_t* secondary, float* profiles, uint32_t nthreads, int mix);
void tracyhost_synth_decompose_batch(uint64_t seed0, uint32_t nt, uint32_t n, uint32_t mf, uint8_t* refs, int32_t* signal,
                                     int32_t* bcpos, uint8_t* primary, uint8_t* secondary, float* profiles, uint32_t nthreads) {
  synth_decompose_batch_mix(seed0, nt, n, mf, refs, signal, bcpos, primary, secondary, profiles, nthreads, 0);
}
void tracyhost_synth_decompose_batch2(uint64_t seed0, uint32_t nt, uint32_t n, uint32_t mf, uint8_t* refs, int32_t* signal,
                                      int32_t* bcpos, uint8_t* primary, uint8_t* secondary, float* profiles, uint32_t nthreads, int mix) {
  synth_decompose_batch_mix(seed0, nt, n, mf, refs, signal, bcpos, primary, secondary, profiles, nthreads, mix);
}
static void synth_decompose_batch_mix(uint64_t seed0, uint32_t nt, uint32_t n, uint32_t mf, uint8_t* refs, int32_t* signal,
                                      int32_t* bcpos, uint8_t* primary, uint8_t* secondary, float* profiles, uint32_t nthreads, int mix) {
  if (nthreads == 0) nthreads = tracy_amd::usable_threads();
  const uint32_t ns = 12 * mf + 12;
  auto work = [&](uint32_t tid) {
    std::vector<int32_t> pos(mf + 64);
    for (uint32_t i = tid; i < nt; i += nthreads) {
      for (uint64_t attempt = 0;; ++attempt) {
        const uint64_t seed = seed0 + i + attempt * 1000003ull * nt;
        int32_t indel = 0;
        int32_t* sig = signal + (size_t)i * 4 * ns;
        int kind = (i % 5 == 4) ? 1 : 0;
        if (mix >= 1) kind = ((i % 10 == 8) ? 2 : (i % 10 == 9) ? 3 : 0) | ((i & 1) ? 16 : 0);
        if (mix == 2) kind |= 32;  // ... in low-complexity windows
        const uint32_t npos = tracyhost_synth_decompose(seed, n, mf, 30, kind, 0.6, refs + (size_t)i * n, sig, ns,
                                                        pos.data(), &indel);
        Trace tr;
        tr.traceACGT.resize(4);
        for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(sig + (size_t)k * ns, sig + (size_t)(k + 1) * ns);
        tr.basecallpos.assign(pos.begin(), pos.begin() + npos);
        BaseCalls bc;
        basecall(tr, bc, 0.33f);
        if (bc.primary.size() != mf) continue;
        Profile p;
        createProfile(tr, bc, p, 0, 0);
        std::memcpy(bcpos + (size_t)i * mf, bc.bcPos.data(), sizeof(int32_t) * mf);
        std::memcpy(primary + (size_t)i * mf, bc.primary.data(), mf);
        std::memcpy(secondary + (size_t)i * mf, bc.secondary.data(), mf);
        std::memcpy(profiles + (size_t)i * 6 * mf, p.data(), sizeof(float) * 6 * mf);
        break;
      }
    }
  };
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < nthreads; ++t) th.emplace_back(work, t);
  for (auto& t : th) t.join();
}

// threads the batch entry points start when the caller passes nthreads = 0
uint32_t tracyhost_usable_threads() { return tracy_amd::usable_threads(); }

}  // extern "C"
