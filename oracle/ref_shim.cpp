// ref_shim.cpp -- extern "C" doors onto the REAL reference functions of common/ntcoding.cpp.
// Built only by oracle/Makefile target `_ref`, together with /root/reference/common/ntcoding.cpp compiled
// from where it lies (never copied).  Output: oracle/_ref/libntcoding_ref.so (git-ignored, travels to the GPU box).
// This file contains no reference code: it only forwards C-callable wrappers to the C++ symbols
// declared in the reference header common/ntcoding.h:1-9.
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string>
#include "ntcoding.h"

extern "C" {
int ref_GenerateShapePos(const char* shape) { return GenerateShapePos(std::string(shape)); }
int ref_IsTransitionAtPos(int t) { return IsTransitionAtPos(t); }
uint32_t ref_GetKmerIndexAtPos(char* sequence, size_t pos, uint32_t seed_size) {
    return GetKmerIndexAtPos(sequence, pos, seed_size);
}
void ref_RevComp(char* dst, char* src, size_t rc_start, size_t start, size_t len) {
    RevComp(dst, src, rc_start, start, len);
}
}
