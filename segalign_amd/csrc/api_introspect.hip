// api_introspect.hip -- C-ABI, introspection: call statistics, filter / lookup mode, copies of device state (tests, bench).
#include "engine_internal.h"

using namespace sa;

extern "C" {

// ---- introspection --------------------------------------------------------------------------------------------------
void sa_get_last_call_stats(sa_call_stats* o) { *o = t_stats; }
void sa_set_count_examined(int on) { g_count_examined = on != 0; }
int sa_get_filter_mode(void) {  // which X-drop filter kernel the next plain (non repeat-masker) call uses
    if (g_count_examined) return g_fast_filter ? 1 : 0;
    return g_packed_filter ? 3 : g_fast_filter;
}
int sa_get_lookup_mode(void) {  // how device-seeded calls look seeds up on device 0 right now (builds the table if needed)
    if (g_ndev <= 0 || !g_proc_init) return 0;
    DevCtx* dc = g_dev[0];
    check_set_device(dc->dev, "lookup mode");
    if (!(g_td && g_packed_filter && !g_count_examined && dc->ref2.base && ensure_nbr(dc))) return 0;
    return dc->nbr_ctx ? 2 : 1;
}
uint64_t sa_get_neighbourhood_entries(void) { return (g_ndev > 0 && g_dev[0]->nbr_state == 1) ? g_dev[0]->nbr_total : 0; }

uint32_t sa_get_ref_len(void) { return g_ndev ? g_dev[0]->ref.len : 0; }
uint32_t sa_get_num_index(void) { return g_ndev ? g_dev[0]->num_index : 0; }
uint32_t sa_get_index_table_size(void) { return g_ndev ? g_dev[0]->nkeys : 0; }
uint32_t sa_get_query_len(uint32_t buffer) { return buffer < SA_BUFFER_DEPTH ? g_query_len[buffer] : 0; }

static DevCtx* ctx_of(int dev) {
    if (dev < 0 || dev >= g_ndev) {
        fprintf(stderr, "Error: device %d out of range\n", dev);
        exit(11);
    }
    check_set_device(g_dev[dev]->dev, "copy");
    return g_dev[dev];
}
void sa_copy_ref_codes(int dev, uint8_t* dst) {
    DevCtx* dc = ctx_of(dev);
    check_memcpy(hipMemcpy(dst, dc->ref.codes, dc->ref.len, hipMemcpyDeviceToHost), "ref codes");
}
void sa_copy_index_table(int dev, uint32_t* dst) {
    DevCtx* dc = ctx_of(dev);
    check_memcpy(hipMemcpy(dst, dc->bucket_start + 1, (size_t)dc->nkeys * sizeof(uint32_t), hipMemcpyDeviceToHost), "index table");
}
void sa_copy_pos_table(int dev, uint32_t* dst) {
    DevCtx* dc = ctx_of(dev);
    check_memcpy(hipMemcpy(dst, dc->pos_table, (size_t)dc->num_index * sizeof(uint32_t), hipMemcpyDeviceToHost), "pos table");
}
void sa_copy_query_codes(int dev, uint32_t buffer, int rev, uint8_t* dst) {
    DevCtx* dc = ctx_of(dev);
    SeqBuf& b = rev ? dc->query_rc[buffer] : dc->query[buffer];
    check_memcpy(hipMemcpy(dst, b.codes, b.len, hipMemcpyDeviceToHost), "query codes");
}

}  // extern "C"
