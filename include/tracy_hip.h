/*
 * tracy_hip.h -- C ABI of the MI355X-native tracy alignment / deconvolution hot path.
 *
 * The reference (gear-genomics/tracy v0.9.1) has no FFI: its boundary for this path is the set of
 * header-only C++ templates listed below.  Each entry point here is the batched, device-side
 * replacement for one of them; tracy_amd/host/tracy_amd.hpp keeps the reference's per-call C++
 * signatures (batch of one) on top of this ABI, INTEGRATION.md shows the binding a tracy maintainer
 * would add.
 *
 *   reference interface (under /root/reference/src)                         replaced by
 *   -------------------------------------------------------------------     ---------------------------
 *   gotohScore(a1,a2,ac,sc)                gotoh.h:12-68                    tracyhip_gotoh_score
 *   gotoh(a1,a2,align,ac,sc)               gotoh.h:71-174                   tracyhip_gotoh_align
 *   needleScore / needle                   needle.h:12-57 / 59-138          tracyhip_needle_score / _align
 *   _createAlignment (string / profile)    align.h:196-223, 254-293         tracyhip_alignment_rows
 *   DnaScore<int>, AlignConfig<H,V>        align.h:11-32, 37-80             tracyhip_params
 *   sage() hot section                     sage.h:191-311                   tracyhip_align_traces
 *   indigo() hot section                   indigo.h:190-388                 tracyhip_decompose_traces
 *   findBreakpoint                         decompose.h:7-56                 tracyhip_find_breakpoint
 *   findHomozygousBreakpoint               decompose.h:59-128               tracyhip_find_homozygous_breakpoint
 *   decomposeAlleles                       decompose.h:179-376              tracyhip_decompose_alleles
 *   generateSecondaryDecomposed            decompose.h:378-410              tracyhip_secondary_decomposed
 *   allelicFraction                        decompose.h:412-621              tracyhip_allelic_fraction
 *   trimReferenceSlice                     fmindex.h:429-463                tracyhip_trim_reference_slice
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer passed in; nothing is retained after return.
 *   - "metadata on the host, payload where `mem` says": offset / length / index arrays are ALWAYS host
 *     arrays; sequence payloads and result arrays are host pointers (TRACYHIP_MEM_HOST, staged through
 *     the library's device buffers) or device pointers (TRACYHIP_MEM_DEVICE, zero copy).
 *   - every call returns TRACYHIP_OK (0) or a negative error; tracyhip_last_error() gives the text.
 *     The reference's DP functions cannot fail (gotoh.h has no checks); the extra errors here are
 *     bad arguments, HIP failures, out-of-memory and parameter ranges the int32 kernels cannot hold.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute call fails.
 */
#ifndef TRACY_HIP_H
#define TRACY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRACYHIP_OK 0
#define TRACYHIP_ERR_ARG (-1)      /* NULL pointer, inconsistent sizes */
#define TRACYHIP_ERR_HIP (-2)      /* a HIP runtime call failed */
#define TRACYHIP_ERR_OOM (-3)      /* device or host allocation failed */
#define TRACYHIP_ERR_RANGE (-4)    /* scores / lengths outside what the int32 kernels represent */
#define TRACYHIP_ERR_NODEVICE (-5) /* no gfx950 device visible */

#define TRACYHIP_MEM_HOST 0
#define TRACYHIP_MEM_DEVICE 1

/* sequence payload kinds: what TAlign1/TAlign2 is in the reference call */
#define TRACYHIP_SEQ_CHAR 0    /* std::string: raw bytes, scored by byte equality (align.h:96-101) */
#define TRACYHIP_SEQ_PROFILE 1 /* boost::multi_array<float,2>[6][len], p[k][j] at k*len+j (align.h:103-118) */

typedef struct tracyhip_ctx tracyhip_ctx;

/* DnaScore<int> (align.h:11-32; inf is the fixed 1000000) + AlignConfig<hfree,vfree> (align.h:37-80).
 * |match|, |mismatch| <= 30000 and (m + n) * (|go| + |ge| + max(|match|, |mismatch|)) + 10^6 < 2^26 (TRACYHIP_ERR_RANGE beyond: the
 * exact range of the x 32 tagged int32 tracebacks); beyond |1000| the table-driven and banded forms give way to slower exact ones */
typedef struct {
  int32_t match;
  int32_t mismatch;
  int32_t go;
  int32_t ge;
  int32_t hfree; /* THorizontal: horizontal moves are free on the first and last ROW */
  int32_t vfree; /* TVertical:   vertical moves are free on the first and last COLUMN */
} tracyhip_params;

/* a set of sequences packed back to back */
typedef struct {
  int32_t kind;           /* TRACYHIP_SEQ_* */
  const void* data;       /* chars, or floats: sequence s occupies [offset[s], offset[s] + (kind ? 6 : 1) * length[s]) */
  const uint64_t* offset; /* HOST array, element offsets (bytes for CHAR, floats for PROFILE) */
  const uint32_t* length; /* HOST array, number of columns */
  uint32_t count;         /* number of sequences */
} tracyhip_seqset;

/* npairs independent DP problems: pair i aligns a1[a1_index[i]] (rows) with a2[a2_index[i]] (columns) */
typedef struct {
  uint32_t npairs;
  tracyhip_seqset a1;
  tracyhip_seqset a2;
  const uint32_t* a1_index; /* HOST array or NULL (= identity) */
  const uint32_t* a2_index; /* HOST array or NULL (= identity) */
} tracyhip_pairs;

/* ---- lifecycle --------------------------------------------------------------------------------- */
/* Host threads: the per-trace loops the pipelines run between two launches (descriptors, bands, verdicts) use a few worker
   threads of the process -- as many as it may run on, at most 8; TRACYHIP_HOST_THREADS=<n> in the environment sets the number
   (one process per GPU on a shared node: the cores of the node divided by the local ranks). */
int tracyhip_device_count(int* count);
int tracyhip_create(int device, tracyhip_ctx** ctx);
int tracyhip_destroy(tracyhip_ctx* ctx);
/* run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int tracyhip_set_stream(tracyhip_ctx* ctx, void* hip_stream);
/* upper bound for the library's device workspace (traceback planes are chunked to fit); 0 = default */
int tracyhip_set_workspace_limit(tracyhip_ctx* ctx, uint64_t bytes);
/* Lanes (1..8, default 1): tracyhip_align_traces / tracyhip_decompose_traces split a batch into `lanes` contiguous
   chunks and run them concurrently, each on its own stream with its own workspace and host thread.  The chunks'
   kernels fill each other's tails and the host stages between kernels (orientation decision, trimReferenceSlice
   geometry) overlap with device work; results are the same arrays as with one lane.  The calls stay synchronous:
   inputs enqueued on the context's stream are waited for, everything is complete on return.  Needs the per-trace
   result regions (ops_offset) in trace order; otherwise the call runs on one lane. */
int tracyhip_set_lanes(tracyhip_ctx* ctx, uint32_t lanes);
/* waits for the context's stream AND for every *_async call issued on the context; returns the first error one of those
   calls produced since the last synchronize (tracyhip_last_error() then holds its text), TRACYHIP_OK otherwise */
int tracyhip_synchronize(tracyhip_ctx* ctx);
/* Options.  Every switch of the library is read from the environment ONCE, when a context is created (TRACYHIP_<NAME>, e.g.
   TRACYHIP_NO_STREAM=1), and changed afterwards only through this call; name is the variable without the prefix, in any case:
     no_stream (pipelines planned by the host between launches instead of stream-ordered), no_narrow, no_compact, no_screen,
     no_band, no_band16, no_front, no_prefix, no_vote, no_origin, no_subwindow, no_prelim_origin, no_cq, no_fused_walk, no_cont16, no_quads, no_fork, no_decomp_wave, no_af_split, no_front_lists, no_origin_band, sweeps_alone (measurement: the full sweeps of the orientation stage on a device of their own)   "0" / "1"
     band_w  (half width of the certified band of the final alignments; -1 = from the preliminary alignment, 0 = whole matrices)
     ckpt_b  (steps between wavefront checkpoints, 32 .. 1024)      verbose  (one line per pipeline stage on stderr)
     quad_tier_min  (stream-ordered pipelines: traces / alleles from which a pruned sweep gets its narrow first tier; default 32768)
     front_list_min (... from which its later tiers, and the allele prefixes of `tracy decompose`, run over device-side lists of the units
                    that are left instead of skipping the others in place; default 1024)
   Every option selects another EXACT path (A/B measurements, tests of the fallback tiers); none changes a result.  Lanes inherit.
   TRACYHIP_HOST_THREADS, TRACYHIP_HOST_TIMERS, TRACYHIP_LDS_PAD and TRACYHIP_LDS_STAGE_LIMIT are per process (read once).
   tracyhip_describe writes the current settings as "name=value" lines (at most cap - 1 bytes) and returns the length needed. */
int tracyhip_set_option(tracyhip_ctx* ctx, const char* name, const char* value);
int tracyhip_describe(tracyhip_ctx* ctx, char* buf, size_t cap);
/* glibc allocator settings that keep the descriptor vectors of the HOST-PLANNED pipelines (no_stream, and the fallback tiers) on the
   heap: M_MMAP_THRESHOLD 32 MB, M_TRIM_THRESHOLD 1 GB, M_TOP_PAD 64 MB.  Process-wide, therefore opt-in: a command-line tool or a
   benchmark calls it once; a library loaded into somebody else's process does not touch the allocator.  (TRACYHIP_MALLOPT=1 in the
   environment does the same at the first tracyhip_create.) */
int tracyhip_tune_host_allocator(void);
/* Which tiers the traces of the last tracyhip_align_traces / tracyhip_decompose_traces call on this context took (summed over its
   lanes).  The pipelines certify every shortcut per trace and repeat what fails on a wider form: these counters say how often. */
typedef struct {
  uint32_t traces;
  uint32_t stream_ordered;       /* 1: planned on the device, one host synchronisation at the end; 0: planned by the host */
  uint32_t host_syncs;           /* host synchronisations of the call */
  uint32_t fallback_traces;      /* stream-ordered call: traces handed to the host-planned tiers afterwards */
  uint32_t pruned;               /* orientation stage: traces whose voted strand took the pruned sweep (front.h) */
  uint32_t pruned_uncertified;   /* ... of which not certified in either tier (swept in full) */
  uint32_t prelim_banded;        /* preliminary alignment on the band kernels */
  uint32_t prelim_repeated;      /* ... of which repeated on the wider form */
  uint32_t final_banded;         /* `tracy align`: final alignments on a certified band */
  uint32_t final_repeated;
  uint32_t allele_pruned[2];     /* `tracy decompose`: gotoh(allele k, window) by the pruned sweep */
  uint32_t allele_uncertified[2];
  uint32_t allele_banded[3];     /* allele k vs its slice (k = 0, 1), allele 1 vs allele 2 (k = 2) on the band kernels */
  uint32_t allele_repeated[3];
  uint32_t allele_shared_prefix; /* stream-ordered `tracy decompose`: traces whose second allele begins with the same 128 characters as the first and
                                    reads the prefix row kept for it (counted in allele_pruned[1] as well) */
} tracyhip_call_stats;
int tracyhip_last_call_stats(tracyhip_ctx* ctx, tracyhip_call_stats* out);
const char* tracyhip_last_error(void);
const char* tracyhip_version(void);

/* ---- DP ---------------------------------------------------------------------------------------- */
/* gotohScore, gotoh.h:12-68: scores[i] = s[n] of pair i. */
int tracyhip_gotoh_score(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm,
                         int mem, int32_t* scores);
/* gotoh, gotoh.h:71-174.  ops receives the reference's `btr` (gotoh.h:147-167): 's','h','v' in PUSH
 * ORDER, i.e. from the end of the alignment to its start; pair i writes ops_len[i] <= m+n bytes at
 * ops + ops_offset[i] (ops_offset is a HOST array).  scores may be NULL. */
int tracyhip_gotoh_align(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm,
                         int mem, int32_t* scores, uint8_t* ops, const uint64_t* ops_offset,
                         uint32_t* ops_len);
/* gotoh (gotoh.h:71-174) on a diagonal band chosen by the caller: pair i is swept on the diagonals c - r in [band_lo[i], band_hi[i]]
 * only (the last row on to column n: its free trailing run), cells outside read as -inf.  Scores, `btr` (and ends) are those of
 * tracyhip_gotoh_align whenever the band holds every optimal path, which is the caller's to establish -- e.g. for strings with
 * free horizontal end gaps: a path that leaves [-W - (m-n)+, W + (n-m)+] makes more than W vertical gap steps, so it scores at
 * most match*m - (match + |ge|)(W+1) - |go|; a banded score above that is the optimum.  This is the form the
 * pipelines below use internally for their final alignments.  CHAR x CHAR or PROFILE x CHAR pairs, AlignConfig<hfree,false>,
 * go <= 0, ge < 0, bands of at most 184 diagonals and references of at most ~14 800 columns (the codes of four pairs are staged in
 * LDS; TRACYHIP_ERR_RANGE beyond).  CHAR rows must hold the letters A C G T N only (the kernels score through a five-letter table,
 * so any other byte would mismatch an identical column byte where gotoh.h compares bytes): TRACYHIP_ERR_ARG otherwise -- such
 * strings go through tracyhip_gotoh_align.  Columns may hold any byte (other letters mismatch every row, as in gotoh.h).  A pair whose traceback walk leaves its band reports
 * ops_len 0.  ends != NULL: the origin-tracking sweep instead of the traceback -- scores and ends[2i] = leading 'h' columns,
 * ends[2i+1] = last column that is not a trailing 'h' (what trimReferenceSlice, fmindex.h:429-463, reads); ops* may be NULL then. */
int tracyhip_gotoh_banded(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, const int32_t* band_lo,
                          const int32_t* band_hi, int mem, int32_t* scores, uint8_t* ops, const uint64_t* ops_offset, uint32_t* ops_len,
                          uint32_t* ends);
/* needleScore / needle, needle.h:12-57 / 59-138 (linear gap cost ge; profiles scored in double). */
int tracyhip_needle_score(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm,
                          int mem, int32_t* scores);
int tracyhip_needle_align(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm,
                          int mem, int32_t* scores, uint8_t* ops, const uint64_t* ops_offset,
                          uint32_t* ops_len);
/* _createAlignment, align.h:196-223 (CHAR) / 254-293 (PROFILE: consensus characters).  For pair i,
 * row0/row1 receive ops_len[i] bytes each at rows0 + ops_offset[i] / rows1 + ops_offset[i]. */
int tracyhip_alignment_rows(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, int mem, const uint8_t* ops,
                            const uint64_t* ops_offset, const uint32_t* ops_len, uint8_t* rows0,
                            uint8_t* rows1);

/* ---- whole `tracy align` hot section (sage.h:191-311) for a batch of traces ---------------------
 * Per trace t:   full profile  P_f = float[6][mf]        (createProfile(tr,bc), sage.h:193)
 *                trimmed profile = columns [trim_left, mf - trim_right) of P_f (sage.h:197; identical
 *                values: createProfile computes each column independently, profile.h:28-51)
 *                reference window  R = bytes[n]           (loadSingleFasta output, sage.h:226)
 * Steps (all on the device): gotohScore(trim, R) and gotohScore(trim, revcomp R) (sage.h:239-240);
 * forward iff gsFwd > gsRev (:247); gotoh(trim, oriented R) (:258); trimReferenceSlice (:259,
 * fmindex.h:429-463); gotoh(full, trimmed slice) (:311). */
typedef struct {
  uint32_t ntraces;
  tracyhip_seqset profiles; /* kind PROFILE, one full profile per trace */
  tracyhip_seqset refs;     /* kind CHAR */
  const uint32_t* ref_index; /* HOST array or NULL (= identity) */
  uint32_t trim_left;       /* SageConfig.trimLeft  (sage.h:88) */
  uint32_t trim_right;      /* SageConfig.trimRight (sage.h:89) */
  const uint8_t* oriented;  /* HOST array or NULL.  Non-NULL = the indexed-genome path (sage.h:217-221): the caller anchored
                               every trace by k-mer seeding (getReferenceSlice, fmindex.h:236-326) and passes refs that are
                               ALREADY oriented (reverse-complemented for reverse traces); oriented[t] = rs.forward.  No
                               orientation scores are computed: score_fwd and score_rev both receive gotohScore(trim, ref). */
  uint32_t strand_by_certificate; /* 0 (default, what a zero-initialised job gets): both gotohScore calls run in full; score_fwd and
                               score_rev are gsFwd and gsRev.  1 (opt-in): the strand is still decided exactly as the reference
                               does (forward iff gsFwd > gsRev), but the LOSING orientation may be represented by a certified
                               upper bound of its score instead of the score itself (a prefix of its DP suffices to prove that
                               it cannot win; see pipeline.hip) -- fewer cells, same decision, same alignments. */
} tracyhip_align_job;

typedef struct {
  int32_t* score_fwd;     /* [ntraces] gsFwd (with strand_by_certificate: exact for the winning orientation only) */
  int32_t* score_rev;     /* [ntraces] gsRev (idem) */
  uint8_t* forward;       /* [ntraces] 1 = rs.forward */
  int32_t* score_prelim;  /* [ntraces] score of the preliminary alignment (sage.h:258), may be NULL */
  uint32_t* slice_begin;  /* [ntraces] ri: offset of the trimmed slice in the ORIENTED reference */
  uint32_t* slice_len;    /* [ntraces] length of the trimmed slice (after substr clamping) */
  uint32_t* ref_pos;      /* [ntraces] rs.pos after trimReferenceSlice (rs.pos starts at 0, sage.h:244) */
  int32_t* score_final;   /* [ntraces] score of the final alignment (sage.h:311) */
  uint8_t* ops;           /* final alignment, push order, at ops + ops_offset[t] (capacity mf + slice) */
  const uint64_t* ops_offset; /* HOST array */
  uint32_t* ops_len;      /* [ntraces] */
} tracyhip_align_result;

int tracyhip_align_traces(tracyhip_ctx* ctx, const tracyhip_align_job* job, const tracyhip_params* prm,
                          int mem, const tracyhip_align_result* out);

/* ---- allele deconvolution (`tracy decompose`, indigo.h:190-388) ------------------------------------ */
/* TraceBreakpoint, fmindex.h:51-56 */
typedef struct {
  int32_t indelshift;
  int32_t traceleft;
  uint32_t breakpoint;
  float best_diff;
} tracyhip_breakpoint;

/* Trace + BaseCalls of a batch (abif.h:28-57), flattened.  signal/bcpos/primary/secondary are payloads
 * (host or device per `mem`), the offset/length arrays are HOST arrays. */
typedef struct {
  uint32_t ntraces;
  const int32_t* signal;         /* trace t: channels A,C,G,T at signal + signal_offset[t] + k*nsamples[t] */
  const uint64_t* signal_offset;
  const uint32_t* nsamples;
  const int32_t* bcpos;          /* bc.bcPos of trace t at bcpos + bc_offset[t] */
  uint8_t* primary;              /* bc.primary   at primary   + bc_offset[t] (rewritten by decomposeAlleles) */
  uint8_t* secondary;            /* bc.secondary at secondary + bc_offset[t] (rewritten by decomposeAlleles) */
  const uint64_t* bc_offset;
  const uint32_t* bc_len;        /* bc.consensus.size() */
  const int32_t* peaks;          /* OPTIONAL (NULL = built on the device from signal + bcpos).  The peak table: the four channels at every
                                    basecall's peak position, peaks[4 * (bc_offset[t] + i) + k] = traceACGT[k][bcPos[i]] of trace t (payload, host
                                    or device per `mem`).  Every read of the chromatogram on this path is at a peak position
                                    (generateSecondaryDecomposed decompose.h:378-410, allelicFraction decompose.h:445-470; createProfile
                                    profile.h:21-52 on the host), so a caller that has the trace in hand -- the command line, while it basecalls --
                                    passes 16 bytes per basecall instead of the chromatogram (192 KB per 1 kb trace); signal, signal_offset,
                                    nsamples and bcpos may then be NULL. */
} tracyhip_basecalls;

/* the IndigoConfig fields decomposeAlleles reads (indigo.h:16-40; CLI defaults 50, 50, 1000, 5) */
typedef struct {
  int32_t trim_left;
  int32_t trim_right;
  int32_t maxindel;              /* 1 .. 65536; traces must hold fewer than 131072 basecalls (TRACYHIP_ERR_RANGE beyond).  The scan
                                  * tables of decomposeAlleles are LDS resident up to 4096 / 8191 basecalls (two size classes) and live
                                  * in global memory beyond that (slower, same results) */
  int32_t madc;
} tracyhip_decomp_params;

/* what decomposeAlleles printed / chose: kind 0 = an indel shift was applied, 1 = "Complex mutation,
 * decomposition: ins, del, error" (decompose.h:315), 2 = "No InDel detected" (decompose.h:327) */
typedef struct {
  int32_t kind;
  int32_t best_ins;
  int32_t best_del;
  int32_t best_fr;
  uint32_t dcp_n; /* rows written to the decomposition table */
  uint32_t pad;
} tracyhip_decomp_status;

/* findBreakpoint, decompose.h:7-56: one breakpoint per profile of the set. */
int tracyhip_find_breakpoint(tracyhip_ctx* ctx, const tracyhip_seqset* profiles, int mem, tracyhip_breakpoint* out);
/* findHomozygousBreakpoint, decompose.h:59-128, applied to the traces whose bps[t].indelshift == 0
 * (indigo.h:314-317).  status[t]: 1 ok, 0 "No valid alignment", -1 "Alignment too short". */
int tracyhip_find_homozygous_breakpoint(tracyhip_ctx* ctx, uint32_t ntraces, const uint8_t* rows0, const uint8_t* rows1,
                                        const uint64_t* rows_offset, const uint32_t* rows_len, int mem,
                                        tracyhip_breakpoint* bps, int32_t* status);
/* decomposeAlleles, decompose.h:179-376.  rows0/rows1: the 2-row alignment of the trimmed trace vs the
 * reference slice (tracyhip_alignment_rows); bps: payload array; refslice_len: HOST array (rs.refslice.size()).
 * The decomposition table of trace t (pairs indel, error; decompose.h:273-285) goes to dcp_* + dcp_offset[t],
 * capacity 2*maxindel+2 entries. */
int tracyhip_decompose_alleles(tracyhip_ctx* ctx, const tracyhip_basecalls* bc, const uint8_t* rows0, const uint8_t* rows1,
                               const uint64_t* rows_offset, const uint32_t* rows_len, const tracyhip_breakpoint* bps,
                               const uint32_t* refslice_len, const tracyhip_decomp_params* prm, int mem,
                               int32_t* dcp_indel, int32_t* dcp_err, const uint64_t* dcp_offset, tracyhip_decomp_status* status);
/* generateSecondaryDecomposed, decompose.h:378-410: secdecomp + bc_offset[t] receives bc.secDecompose. */
int tracyhip_secondary_decomposed(tracyhip_ctx* ctx, const tracyhip_basecalls* bc, int mem, uint8_t* secdecomp);
/* allelicFraction, decompose.h:412-621: fractions[2t], fractions[2t+1] = the returned pair. */
int tracyhip_allelic_fraction(tracyhip_ctx* ctx, const tracyhip_basecalls* bc, const uint8_t* secdecomp, uint32_t trim_left,
                              uint32_t trim_right, int mem, double* fractions);

/* trimReferenceSlice, fmindex.h:429-463, on the two rows of an alignment of a trace (row 0) against its reference slice
 * (row 1): slice_begin[t] = ri after the trimLeft widening, slice_len[t] = the length rs.refslice.substr(ri, risize) has,
 * ref_pos[t] = what the call adds to rs.pos (ri forward; oldlen - ri - risize reverse, 0 when that is negative -- the
 * reference only warns).  refslice_len (rs.refslice.size()) and forward (rs.forward) are HOST arrays. */
int tracyhip_trim_reference_slice(tracyhip_ctx* ctx, uint32_t ntraces, const uint8_t* rows0, const uint8_t* rows1,
                                  const uint64_t* rows_offset, const uint32_t* rows_len, const uint32_t* refslice_len,
                                  const uint8_t* forward, uint32_t trim_left, uint32_t trim_right, int mem,
                                  uint32_t* slice_begin, uint32_t* slice_len, uint32_t* ref_pos);

/* ---- whole `tracy decompose` hot section (indigo.h:190-388) for a batch of traces, FASTA reference ----
 * findBreakpoint(trimmed profile) -> orientation scores -> gotoh(trimmed, oriented reference) with the
 * score gate of indigo.h:303-309 -> findHomozygousBreakpoint when no shift was seen -> decomposeAlleles ->
 * generateSecondaryDecomposed -> allelicFraction -> per allele: gotoh(seq, rs.refslice), trimReferenceSlice,
 * gotoh(seq, trimmed slice) -> gotoh(primary, secondary) global.  bc.primary / bc.secondary are rewritten
 * in place (decomposed basecalls).  prm->hfree/vfree are ignored (the configs are fixed by indigo.h). */
typedef struct {
  uint32_t ntraces;
  tracyhip_seqset profiles;      /* kind PROFILE: createProfile(tr, bc) of every trace, bc_len[t] columns */
  tracyhip_basecalls bc;
  tracyhip_seqset refs;          /* kind CHAR, upper-case [ACGTN] */
  const uint32_t* ref_index;     /* HOST array or NULL */
  tracyhip_decomp_params dprm;
  const uint8_t* oriented;       /* HOST array or NULL; as in tracyhip_align_job: refs already oriented by k-mer seeding
                                    (indigo.h:213-218), oriented[t] = rs.forward; score_fwd / score_rev are then zero */
  tracyhip_seqset ref_profiles;  /* data NULL = unused.  Wildtype-trace reference (indigo.h:249-289): profile of the wildtype
                                    trace, oriented by the caller (needs `oriented`), parallel to refs, which then hold the
                                    wildtype's (oriented) primary basecalls = rs.refslice */
  uint32_t strand_by_certificate; /* as in tracyhip_align_job: 0 (default) = both orientation scores exact; 1 = the losing
                                    orientation's score may be a certified upper bound */
} tracyhip_decompose_job;

typedef struct {
  tracyhip_breakpoint* bp;       /* [ntraces] breakpoint used by decomposeAlleles */
  int32_t* status;               /* [ntraces] 0 ok; -1 "Alignment of trace to reference failed!" (indigo.h:306-309);
                                    findHomozygousBreakpoint failed (:316): -2 "No valid alignment found ..." (decompose.h:81),
                                    -3 "Alignment too short ..." (:92).  Later outputs of such traces are unspecified. */
  int32_t* score_fwd;
  int32_t* score_rev;
  uint8_t* forward;
  int32_t* score_trim;           /* aliTrimScore (indigo.h:302) */
  int32_t* dcp_indel;            /* decomposition table, trace t at dcp_offset[t], capacity 2*maxindel+2 */
  int32_t* dcp_err;
  const uint64_t* dcp_offset;    /* HOST array */
  tracyhip_decomp_status* dstatus;
  uint8_t* secdecomp;            /* bc.secDecompose, trace t at bc.bc_offset[t] */
  double* fractions;             /* [2*ntraces] allelicFraction */
  /* allele alignments k = 0 (primary vs its trimmed slice), 1 (secondary), 2 (primary vs secondary, global);
   * ops in push order, capacity len(seq) + len(reference) (k < 2) or 2*len(seq) (k = 2) */
  uint32_t* slice_begin[2];
  uint32_t* slice_len[2];
  uint32_t* ref_pos[2];
  int32_t* score[3];
  uint8_t* ops[3];
  const uint64_t* ops_offset[3]; /* HOST arrays */
  uint32_t* ops_len[3];
} tracyhip_decompose_result;

int tracyhip_decompose_traces(tracyhip_ctx* ctx, const tracyhip_decompose_job* job, const tracyhip_params* prm, int mem,
                              const tracyhip_decompose_result* out);

/* ---- asynchronous forms (SURVEY.md 8b "Threading": synchronous by default with an async variant) ---------------------
 * Same arguments and results as the call without the suffix; the call returns as soon as the work is queued on the
 * context.  A context executes its calls in issue order on its own worker thread and stream (the pipelines need the
 * host between kernels -- orientation decision, trimReferenceSlice geometry -- so "enqueue" means this queue, not only
 * the HIP stream).  The structs are copied; every array they point to (inputs, offsets, results) must stay valid and
 * untouched until tracyhip_synchronize(ctx) returns, which is also when results are final and errors are reported.
 * A synchronous call on a context with queued work waits for that work first.  Several contexts (one per host thread,
 * or the members of a group) run concurrently. */
int tracyhip_gotoh_score_async(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, int mem, int32_t* scores);
int tracyhip_gotoh_align_async(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, int mem, int32_t* scores,
                               uint8_t* ops, const uint64_t* ops_offset, uint32_t* ops_len);
int tracyhip_align_traces_async(tracyhip_ctx* ctx, const tracyhip_align_job* job, const tracyhip_params* prm, int mem,
                                const tracyhip_align_result* out);
int tracyhip_decompose_traces_async(tracyhip_ctx* ctx, const tracyhip_decompose_job* job, const tracyhip_params* prm, int mem,
                                    const tracyhip_decompose_result* out);

/* ---- device groups: the GPUs of one node behind one handle (north star: "batches of traces shard embarrassingly across
 * the 8 GPUs of one node") ------------------------------------------------------------------------------------------
 * One context per device, one host thread per context.  A batch call on a group cuts the batch into contiguous blocks
 * of traces (tracyhip_group_gotoh_score: the pair list into slices of equal DP cell count, the sequence sets replicated --
 * the all-pairs matrix of msa.h:33-42), runs each block through its device and returns when all are complete; results
 * land in the caller's arrays exactly as from the single-device call.  HOST buffers only (every block stages its own part
 * through its own device; nothing crosses devices).  devices == NULL: the first `ndevices` visible devices (0 = all).
 * A device may be listed more than once (two contexts on one GPU).  Multi-PROCESS jobs use one plain context per rank
 * and gather over RCCL instead (tracy_amd/shard.py, bench.py). */
typedef struct tracyhip_group tracyhip_group;
int tracyhip_group_create(const int* devices, int ndevices, tracyhip_group** group);
int tracyhip_group_destroy(tracyhip_group* group);
int tracyhip_group_size(const tracyhip_group* group);
tracyhip_ctx* tracyhip_group_context(tracyhip_group* group, int i); /* member i, e.g. for tracyhip_set_workspace_limit */
int tracyhip_group_set_lanes(tracyhip_group* group, uint32_t lanes);
int tracyhip_group_gotoh_score(tracyhip_group* group, const tracyhip_pairs* pairs, const tracyhip_params* prm, int32_t* scores);
int tracyhip_group_align_traces(tracyhip_group* group, const tracyhip_align_job* job, const tracyhip_params* prm,
                                const tracyhip_align_result* out);
int tracyhip_group_decompose_traces(tracyhip_group* group, const tracyhip_decompose_job* job, const tracyhip_params* prm,
                                    const tracyhip_decompose_result* out);
/* bounds[0 .. parts] of `parts` contiguous slices of the pair list with (nearly) equal DP cell count; host arithmetic, no device */
int tracyhip_pair_bounds(const tracyhip_pairs* pairs, uint32_t parts, uint64_t* bounds);

/* ---- result compaction for the final gather of a sharded job (no counterpart in the reference, which is one process) ----
 * The pipelines write variable-length results (traceback strings, decomposition tables) into fixed-capacity regions the caller lays out
 * (ops_offset[t] = t * capacity, dcp_offset[t] = t * (2 maxindel + 2)).  Before a rank ships them to the rank that collects the job's
 * results it packs the used parts back to back: region i of a payload kind = bytes [i * stride_bytes, (i + 1) * stride_bytes) of src, of
 * which the first lens[i * lens_stride] * elem_bytes bytes are used (never more than the stride).  src, lens and dst are DEVICE pointers;
 * lens_stride lets a field of a record array serve (tracyhip_decomp_status::dcp_n: lens = &dstatus[0].dcp_n, lens_stride = 6).
 * tracyhip_pack_ragged_multi packs up to 16 kinds in one pass, kind-major (all regions of kind 0 in region order, then kind 1, ...) --
 * one scan, one copy launch, one synchronisation; kind_bytes[k] (HOST) receives the packed size of kind k.  dst == NULL: sizes only.
 * TRACYHIP_ERR_ARG when the total exceeds dst_cap (nothing is written then; with dst_cap >= the sum of n * stride_bytes the copy is
 * queued without waiting for the sizes).  Synchronous. */
typedef struct {
  const void* src;
  uint64_t stride_bytes;
  uint32_t elem_bytes;
  const uint32_t* lens;
  uint32_t lens_stride;
} tracyhip_ragged_src;
int tracyhip_pack_ragged_multi(tracyhip_ctx* ctx, const tracyhip_ragged_src* kinds, uint32_t nkinds, uint32_t n, void* dst, uint64_t dst_cap,
                               uint64_t* kind_bytes);
int tracyhip_pack_ragged(tracyhip_ctx* ctx, const void* src, uint64_t stride_bytes, uint32_t elem_bytes, const uint32_t* lens, uint32_t lens_stride,
                         uint32_t n, void* dst, uint64_t dst_cap, uint64_t* total_bytes);

/* ---- kernel timing (HIP events recorded on the context's stream around each DP / walker launch) ---
 * The reference has only the optional gperftools wrapper (sage.h:60-62); this is the hook bench.py uses
 * for its roofline line.  bytes = ALGORITHMIC bytes of the launches: 0.5 B per traceback cell + the
 * inputs once (1 B per reference base, 24 B per profile column, 1 B per string base) + 4 B score. */
#define TRACYHIP_TIMER_SCORE 0 /* score-only DP kernels */
#define TRACYHIP_TIMER_TRACE 1 /* traceback DP kernels  */
#define TRACYHIP_TIMER_WALK 2  /* traceback walkers     */
#define TRACYHIP_TIMER_BAND 3  /* band tracebacks (checkpointed traceback of the align / decompose pipelines); cells = the cells the
                                  kernel really re-swept (strip height x steps of the bands its paths cross), not m x n */
#define TRACYHIP_TIMER_PREFIX 4 /* prefix-bound kernels (strand by certificate); cells = rows actually swept x columns */
#define TRACYHIP_TIMER_ORIGIN 5 /* origin-tracking sweeps (gotoh() whose alignment only trimReferenceSlice reads) */
#define TRACYHIP_TIMER_DECOMP 6 /* decomposeAlleles kernel; cells = alignment columns, bytes = rows + basecalls + table */
#define TRACYHIP_TIMER_AFRAC 7  /* allelicFraction kernel; cells = grid points x diffnuc bound, bytes = signal windows read */
#define TRACYHIP_TIMER_MISC 8   /* findBreakpoint, findHomozygousBreakpoint, generateSecondaryDecomposed, alignment rows, trims */
#define TRACYHIP_TIMER_FRONT 9  /* pruned orientation sweep of `tracy align` (front.h): band placement, the band sweep below the prefix
                                   rows, certificate; cells = the band's, bytes = tables + codes + the kept row (read twice) */
#define TRACYHIP_TIMER_COUNT 10
typedef struct {
  double ms;         /* summed launch durations */
  uint64_t launches;
  uint64_t cells;    /* DP cells (m*n summed over pairs) */
  uint64_t bytes;    /* algorithmic HBM bytes */
} tracyhip_kernel_timing;
int tracyhip_timing_enable(tracyhip_ctx* ctx, int on);
int tracyhip_timing_reset(tracyhip_ctx* ctx);
int tracyhip_timing_get(tracyhip_ctx* ctx, int which, tracyhip_kernel_timing* out);

#ifdef __cplusplus
}
#endif
#endif
