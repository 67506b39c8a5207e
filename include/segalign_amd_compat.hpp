// segalign_amd_compat.hpp -- the reference's OWN engine symbols, rebuilt on the C-ABI of libsegalign_hip.so.
//
// Drop-in recipe for a SegAlign maintainer (details in INTEGRATION.md):
//   1. remove common/seed_filter_interface.cu, common/seed_pos_table.cu and src/seed_filter.cu from add_executable()
//      in CMakeLists.txt:25-34 (they are what this engine replaces);
//   2. add ONE translation unit that does
//          #define SEGALIGN_AMD_COMPAT_DEFINE
//          #include "segalign_amd_compat.hpp"
//      and link with -lsegalign_hip.
// src/main.cpp, src/seeder.cpp, src/segment_printer.cpp and common/ntcoding.cpp stay untouched: they keep including
// their own seed_filter.h / seed_filter_interface.h / ntcoding.h, whose declarations the definitions below satisfy:
//
//   symbol (reference declaration)                                   defined originally at
//   InitializeInterface_ptr   g_InitializeInterface   (seed_filter_interface.h:3,8)   seed_filter_interface.cu:115
//   SendRefWriteRequest_ptr   g_SendRefWriteRequest   (seed_filter_interface.h:4,9)   seed_filter_interface.cu:116
//   ClearRef_ptr              g_ClearRef              (seed_filter_interface.h:5,10)  seed_filter_interface.cu:117
//   InitializeProcessor_ptr   g_InitializeProcessor   (src/seed_filter.h:4,10)        src/seed_filter.cu:942
//   SendQueryWriteRequest_ptr g_SendQueryWriteRequest (src/seed_filter.h:5,11)        src/seed_filter.cu:943
//   SeedAndFilter_ptr         g_SeedAndFilter         (src/seed_filter.h:6,12)        src/seed_filter.cu:944
//   ClearQuery_ptr            g_ClearQuery            (src/seed_filter.h:7,13)        src/seed_filter.cu:945
//   ShutdownProcessor_ptr     g_ShutdownProcessor     (src/seed_filter.h:8,14)        src/seed_filter.cu:946
//   void GenerateSeedPosTable(char*, size_t, uint32_t, uint32_t, int, int) (ntcoding.h:9)  seed_pos_table.cu:49
//
// The repeat-masker binary (repeat_masker_src/) declares THE SAME NAMES with different signatures
// (repeat_masker_src/seed_filter.h:4-14): its translation unit does instead
//          #define SEGALIGN_AMD_COMPAT_DEFINE_RM
//          #include "segalign_amd_compat.hpp"
// which defines
//   InitializeProcessor_ptr   g_InitializeProcessor   (repeat_masker_src/seed_filter.h:4,10)   repeat_masker_src/seed_filter.cu:983
//   SendQueryWriteRequest_ptr g_SendQueryWriteRequest  void(*)()                                (:5,11)   :984
//   SeedAndFilter_ptr         g_SeedAndFilter          vector<segmentPair>(*)(vector<uint64_t>, bool rev, uint32_t ref_start,
//                                                      uint32_t ref_end)                        (:6,12)   :985
//   ClearQuery_ptr            g_ClearQuery             void(*)()                                (:7,13)   :986
//   ShutdownProcessor_ptr     g_ShutdownProcessor                                               (:8,14)   :987
// plus the three common/ symbols and GenerateSeedPosTable as above.
//
// Two things the reference keeps in host globals are bridged explicitly:
//   * the seed shape: GenerateShapePos (ntcoding.cpp:21-37) fills `shape_pos[] / transition_pos[]`, which
//     ntcoding.cpp exports as plain globals (ntcoding.cpp:6-8); the compat GenerateSeedPosTable re-derives the shape
//     string from them and hands it to sa_generate_shape_pos() before the device build;
//   * the query arena: SendQueryWriteRequest reads `query_DRAM->buffer` (src/store.h:7, seed_filter.cu:910); the
//     compat wrapper passes that pointer to the C entry point.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "segalign_amd.h"

// ---- the reference's boundary types (src/graph.h:25-30, src/seed_filter.h:4-8, seed_filter_interface.h:3-6) --------
#ifndef SEGALIGN_AMD_COMPAT_NO_TYPES
struct segmentPair {
    uint32_t ref_start;
    uint32_t query_start;
    uint32_t len;
    int score;
};
typedef int (*InitializeInterface_ptr)(int num_gpu);
typedef void (*SendRefWriteRequest_ptr)(char* seq, size_t addr, uint32_t len);
typedef void (*ClearRef_ptr)();
typedef void (*ShutdownProcessor_ptr)();
typedef void (*InitializeProcessor_ptr)(bool transition, uint32_t WGA_CHUNK, uint32_t input_seed_size, int* sub_mat,
                                        int input_xdrop, int input_hspthresh, bool input_noentropy);
#if defined(SEGALIGN_AMD_COMPAT_RM) || defined(SEGALIGN_AMD_COMPAT_DEFINE_RM)  // repeat_masker_src/seed_filter.h:5-7
typedef void (*SendQueryWriteRequest_ptr)();
typedef std::vector<segmentPair> (*SeedAndFilter_ptr)(std::vector<uint64_t> seed_offset_vector, bool rev, uint32_t ref_start,
                                                      uint32_t ref_end);
typedef void (*ClearQuery_ptr)();
#else                                                                            // src/seed_filter.h:5-7
typedef void (*SendQueryWriteRequest_ptr)(size_t addr, uint32_t len, uint32_t buffer);
typedef std::vector<segmentPair> (*SeedAndFilter_ptr)(std::vector<uint64_t> seed_offset_vector, bool rev, uint32_t buffer);
typedef void (*ClearQuery_ptr)(uint32_t buffer);
#endif
#endif

static_assert(sizeof(segmentPair) == sizeof(sa_segment_pair), "segmentPair layout (src/graph.h:25-30)");

// Flavour guard.  The two binaries declare the SAME g_* names with different signatures, selected above per translation unit; a TU
// of the repeat-masker binary that includes this header without SEGALIGN_AMD_COMPAT_RM would see the src/ signatures for the same
// symbols -- a silent type mismatch.  Every TU therefore references a symbol named after ITS flavour, and only the TU that defines
// the pointers (SEGALIGN_AMD_COMPAT_DEFINE / _DEFINE_RM) defines the one of its own flavour: mixed inclusion fails at link time
// with "undefined reference to segalign_amd_compat_flavour_...".
#if defined(SEGALIGN_AMD_COMPAT_RM) || defined(SEGALIGN_AMD_COMPAT_DEFINE_RM)
extern "C" const int segalign_amd_compat_flavour_repeat_masker;
namespace { __attribute__((used)) const int* const segalign_amd_compat_flavour_ref = &segalign_amd_compat_flavour_repeat_masker; }
#else
extern "C" const int segalign_amd_compat_flavour_src;
namespace { __attribute__((used)) const int* const segalign_amd_compat_flavour_ref = &segalign_amd_compat_flavour_src; }
#endif

namespace segalign_amd_compat {

// Where SendQueryWriteRequest finds the query arena.  The reference reads the global `query_DRAM->buffer`
// (src/store.h:7); a host that links this header sets the pointer once after loading the query
// (or defines SEGALIGN_AMD_COMPAT_USE_QUERY_DRAM to read the reference global directly).
inline char*& query_arena() {
    static char* p = nullptr;
    return p;
}

inline int InitializeInterface(int num_gpu) { return sa_initialize_interface(num_gpu); }
inline void SendRefWriteRequest(char* seq, size_t addr, uint32_t len) { sa_send_ref_write_request(seq, addr, len); }
inline void ClearRef() { sa_clear_ref(); }
inline void ShutdownProcessor() { sa_shutdown_processor(); }
inline void InitializeProcessor(bool transition, uint32_t wga_chunk, uint32_t seed_size, int* sub_mat, int xdrop, int hspthresh,
                                bool noentropy) {
    sa_initialize_processor(transition ? 1 : 0, wga_chunk, seed_size, sub_mat, xdrop, hspthresh, noentropy ? 1 : 0);
}
inline void ClearQuery(uint32_t buffer) { sa_clear_query(buffer); }

#ifdef SEGALIGN_AMD_COMPAT_USE_QUERY_DRAM
}  // namespace segalign_amd_compat
#include "store.h"  // the reference's own header: extern DRAM* query_DRAM (src/store.h:7)
namespace segalign_amd_compat {
inline void SendQueryWriteRequest(size_t addr, uint32_t len, uint32_t buffer) {
    sa_send_query_write_request(query_DRAM->buffer, addr, len, buffer);  // src/seed_filter.cu:910
}
#else
inline void SendQueryWriteRequest(size_t addr, uint32_t len, uint32_t buffer) {
    sa_send_query_write_request(query_arena(), addr, len, buffer);
}
#endif

// g_SeedAndFilter: std::vector by value in, std::vector out, element 0 = header (src/seed_filter.cu:682-828)
inline std::vector<segmentPair> SeedAndFilter(std::vector<uint64_t> seed_offset_vector, bool rev, uint32_t buffer) {
    sa_segment_pair* out = nullptr;
    size_t n = sa_seed_and_filter(seed_offset_vector.data(), seed_offset_vector.size(), rev ? 1 : 0, buffer, &out);
    std::vector<segmentPair> v(n);
    for (size_t i = 0; i < n; i++) {
        v[i].ref_start = out[i].ref_start;
        v[i].query_start = out[i].query_start;
        v[i].len = out[i].len;
        v[i].score = out[i].score;
    }
    sa_free_segments(out);
    return v;
}

// repeat masker flavour (repeat_masker_src/seed_filter.h:5-7): the query IS the resident target
inline void RmSendQueryWriteRequest() { sa_rm_send_query_write_request(); }  // repeat_masker_src/seed_filter.cu:951-961
inline void RmClearQuery() { sa_rm_clear_query(); }                          // repeat_masker_src/seed_filter.cu:964-972
// SeedAndFilter(seeds, rev, ref_start, ref_end) (repeat_masker_src/seed_filter.cu:724-876); element 0 packs the 64-bit
// hit / anchor counts as {ref_start, query_start} / {len, score} (:857-861)
inline std::vector<segmentPair> RmSeedAndFilter(std::vector<uint64_t> seed_offset_vector, bool rev, uint32_t ref_start,
                                                uint32_t ref_end) {
    sa_segment_pair* out = nullptr;
    size_t n = sa_rm_seed_and_filter(seed_offset_vector.data(), seed_offset_vector.size(), rev ? 1 : 0, ref_start, ref_end, &out);
    std::vector<segmentPair> v(n);
    for (size_t i = 0; i < n; i++) {
        v[i].ref_start = out[i].ref_start;
        v[i].query_start = out[i].query_start;
        v[i].len = out[i].len;
        v[i].score = out[i].score;
    }
    sa_free_segments(out);
    return v;
}

// shape string ('T' = care + transition, '1' = care, '0' = don't care) from the arrays GenerateShapePos filled
inline std::string shape_from_arrays(const int* shape_pos, int weight, const int* transition_pos, int span) {
    std::string s((size_t)span, '0');
    for (int j = 0; j < weight; j++) s[(size_t)shape_pos[j]] = transition_pos[j] ? 'T' : '1';
    return s;
}

}  // namespace segalign_amd_compat

#if defined(SEGALIGN_AMD_COMPAT_DEFINE) && defined(SEGALIGN_AMD_COMPAT_DEFINE_RM)
#error "one binary links either src/ (SEGALIGN_AMD_COMPAT_DEFINE) or repeat_masker_src/ (SEGALIGN_AMD_COMPAT_DEFINE_RM)"
#endif

#ifdef SEGALIGN_AMD_COMPAT_DEFINE_RM
// ---- the definitions repeat_masker_src/seed_filter.cu:983-987 and common/seed_filter_interface.cu:115-117 used to provide ----
InitializeInterface_ptr g_InitializeInterface = segalign_amd_compat::InitializeInterface;      // seed_filter_interface.cu:115
SendRefWriteRequest_ptr g_SendRefWriteRequest = segalign_amd_compat::SendRefWriteRequest;      // :116
ClearRef_ptr g_ClearRef = segalign_amd_compat::ClearRef;                                       // :117
InitializeProcessor_ptr g_InitializeProcessor = segalign_amd_compat::InitializeProcessor;      // repeat_masker_src/seed_filter.cu:983
SendQueryWriteRequest_ptr g_SendQueryWriteRequest = segalign_amd_compat::RmSendQueryWriteRequest;  // :984
SeedAndFilter_ptr g_SeedAndFilter = segalign_amd_compat::RmSeedAndFilter;                      // :985
ClearQuery_ptr g_ClearQuery = segalign_amd_compat::RmClearQuery;                               // :986
ShutdownProcessor_ptr g_ShutdownProcessor = segalign_amd_compat::ShutdownProcessor;            // :987
extern "C" const int segalign_amd_compat_flavour_repeat_masker = 1;                           // (flavour guard, see above)
#endif

#ifdef SEGALIGN_AMD_COMPAT_DEFINE
// ---- the definitions the reference's .cu files used to provide -----------------------------------------------------
InitializeInterface_ptr g_InitializeInterface = segalign_amd_compat::InitializeInterface;      // seed_filter_interface.cu:115
SendRefWriteRequest_ptr g_SendRefWriteRequest = segalign_amd_compat::SendRefWriteRequest;      // :116
ClearRef_ptr g_ClearRef = segalign_amd_compat::ClearRef;                                       // :117
InitializeProcessor_ptr g_InitializeProcessor = segalign_amd_compat::InitializeProcessor;      // src/seed_filter.cu:942
SendQueryWriteRequest_ptr g_SendQueryWriteRequest = segalign_amd_compat::SendQueryWriteRequest;  // :943
SeedAndFilter_ptr g_SeedAndFilter = segalign_amd_compat::SeedAndFilter;                        // :944
ClearQuery_ptr g_ClearQuery = segalign_amd_compat::ClearQuery;                                 // :945
ShutdownProcessor_ptr g_ShutdownProcessor = segalign_amd_compat::ShutdownProcessor;            // :946
extern "C" const int segalign_amd_compat_flavour_src = 1;                                     // (flavour guard, see above)
#endif

#if defined(SEGALIGN_AMD_COMPAT_DEFINE) || defined(SEGALIGN_AMD_COMPAT_DEFINE_RM)
// globals of common/ntcoding.cpp:6-8 (that file stays in the build)
extern int shape_pos[32];
extern int shape_size;
extern int transition_pos[32];

// common/ntcoding.h:9 ; replaces common/seed_pos_table.cu:49-109
void GenerateSeedPosTable(char* ref_str, size_t start_addr, uint32_t ref_length, uint32_t step, int shape_span, int kmer_size) {
    const std::string shape = segalign_amd_compat::shape_from_arrays(shape_pos, shape_size, transition_pos, shape_span);
    sa_generate_shape_pos(shape.c_str());
    sa_generate_seed_pos_table(ref_str, start_addr, ref_length, step, shape_span, kmer_size);
}
#endif  // SEGALIGN_AMD_COMPAT_DEFINE / _RM
