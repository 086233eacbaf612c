// api_prof.hip -- bpe_prof_*.
// Part of bpe_api.hip, which includes the parts in order (one translation unit).

extern "C" {

int bpe_prof_reset(bpe_ctx *c) {
    if (!c) return BPE_E_ARG;
    TRY(prof_drain(c));
    for (int k = 0; k < BPE_PROF_NKINDS; k++) {
        c->prof_ms[k] = 0;
        c->prof_launches[k] = 0;
        c->prof_bytes[k] = 0;
    }
    return BPE_OK;
}

int bpe_prof_read(bpe_ctx *c, double *ms, uint64_t *launches, uint64_t *alg_bytes) {
    if (!c) return BPE_E_ARG;
    TRY(prof_drain(c));
    for (int k = 0; k < BPE_PROF_NKINDS; k++) {
        if (ms) ms[k] = c->prof_ms[k];
        if (launches) launches[k] = c->prof_launches[k];
        if (alg_bytes) alg_bytes[k] = c->prof_bytes[k];
    }
    return BPE_OK;
}

int bpe_train_stats(bpe_ctx *c, uint64_t *out4) {
    if (!c || !out4) return BPE_E_ARG;
    out4[0] = c->n_dense;
    out4[1] = c->n_sparse;
    out4[2] = c->n_index_builds;
    out4[3] = c->slot_T;
    return BPE_OK;
}

int bpe_train_stats_ex(bpe_ctx *c, uint64_t *out, int n) {
    if (!c || !out || n < 0) return BPE_E_ARG;
    uint64_t chained = 0;
    if (n > 6 && c->d_st) {  // (device-side count: iterations whose pair came off a chain, k_lean.hip)
        DevState stt;
        TRY(read_state(c, &stt));
        chained = stt.chain_taken;
    }
    const uint64_t v[11] = {c->n_dense, c->n_sparse, c->n_index_builds, c->slot_T, c->n_lean, c->n_deferred,
                            chained + c->n_chained, c->n_steps, c->n_full, (uint64_t)c->ts, c->n_fused};
    for (int i = 0; i < n && i < 11; i++) out[i] = v[i];
    return BPE_OK;
}

}  // extern "C"
