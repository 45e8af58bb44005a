// engine.hip -- the policy engine: owns the layer program of one actor-critic (encoder -> recurrent
// state encoder -> heads), the layout of the flat parameter / gradient arenas, and the activation
// workspace; runs `act`, `evaluate` (forward, activations kept) and `backward` as sequences of the
// kernels in this library on one HIP stream.  Python only makes one C call per policy invocation.
//
// Replaces (habitat-baselines/habitat_baselines/): rl/ppo/policy.py:324-402 (NetPolicy.act /
// get_value / evaluate_actions) with PointNavBaselineNet :494-589, and the autograd backward that
// total_loss.backward() (rl/ppo/ppo.py:253) runs through them.
#include <string.h>

#include <algorithm>

#include <string>
#include <vector>

#include "engine.h"

using namespace hab;

static int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

static int build_baseline(hab_policy* e) {
    const hab_policy_desc& d = e->d;
    e->Cin = (d.has_rgb ? 3 : 0) + (d.has_depth ? 1 : 0);
    const bool blind = e->Cin == 0;  // SimpleCNN.is_blind (simple_cnn.py:54,95-97): no visual sensor -> the net is goal -> GRU -> heads
    if (blind && d.goal_dim <= 0) return HAB_ERR_ARG;
    if (d.rnn_type != HAB_RNN_GRU && d.rnn_type != HAB_RNN_LSTM) return HAB_ERR_ARG;
    const int H = d.hidden;
    e->G_ = d.rnn_type == HAB_RNN_GRU ? 3 : 4;
    e->L = d.rnn_layers;
    // SimpleCNN (rl/models/simple_cnn.py:68-93)
    e->c1 = ConvDesc{0, d.H, d.W, blind ? 4 : e->Cin, 32, 8, 8, 4, 0};
    e->c2 = ConvDesc{0, conv_out(d.H, 8, 4, 0), conv_out(d.W, 8, 4, 0), 32, 64, 4, 4, 2, 0};
    e->c3 = ConvDesc{0, e->c2.Ho(), e->c2.Wo(), 64, 32, 3, 3, 1, 0};
    const int h3 = e->c3.Ho(), w3 = e->c3.Wo();
    if (!blind && (h3 <= 0 || w3 <= 0)) return HAB_ERR_ARG;
    e->fc_in = blind ? 0 : 32 * h3 * w3;
    e->rnn_in = (blind ? 0 : H) + d.goal_dim;  // policy.py:532-535
    e->rnn_ld = (e->rnn_in + 15) & ~15;  // rows padded (with zeros) to a multiple of 16: the K-chunk of the fused step projection (rnn.hip)
    const std::string ve = "net.visual_encoder.cnn.";
    if (!blind) {
        e->i_c1w = add_param(e, ve + "0.weight", {32, e->Cin, 8, 8});
        e->i_c1b = add_param(e, ve + "0.bias", {32});
        e->i_c2w = add_param(e, ve + "2.weight", {64, 32, 4, 4});
        e->i_c2b = add_param(e, ve + "2.bias", {64});
        e->i_c3w = add_param(e, ve + "4.weight", {32, 64, 3, 3});
        e->i_c3b = add_param(e, ve + "4.bias", {32});
        e->i_fcw = add_param(e, ve + "6.weight", {H, e->fc_in});
        e->i_fcb = add_param(e, ve + "6.bias", {H});
    } else {
        e->i_c1w = e->i_c1b = e->i_c2w = e->i_c2b = e->i_c3w = e->i_c3b = e->i_fcw = e->i_fcb = -1;
    }
    const std::string rn = "net.state_encoder.rnn.";
    for (int l = 0; l < d.rnn_layers; ++l) {
        const int in = l == 0 ? e->rnn_in : H;
        const std::string sfx = "_l" + std::to_string(l);
        e->i_wih.push_back(add_param(e, rn + "weight_ih" + sfx, {e->G_ * H, in}));
        e->i_whh.push_back(add_param(e, rn + "weight_hh" + sfx, {e->G_ * H, H}));
        e->i_bih.push_back(add_param(e, rn + "bias_ih" + sfx, {e->G_ * H}));
        e->i_bhh.push_back(add_param(e, rn + "bias_hh" + sfx, {e->G_ * H}));
    }
    e->i_aw = add_param(e, "action_distribution.linear.weight", {d.num_actions, H});
    e->i_ab = add_param(e, "action_distribution.linear.bias", {d.num_actions});
    e->i_cw = add_param(e, "critic.fc.weight", {1, H});
    e->i_cb = add_param(e, "critic.fc.bias", {1});

    Arena pk;
    e->pk_c1f = pk.take(32 * 64 * (blind ? 4 : e->Cin));
    e->pk_c1img = (!blind && e->Cin == 4) ? pk.take(obs_conv_weight_image_floats()) : -1;
    e->pk_c2f = pk.take(64 * 16 * 32);
    e->pk_c2d = pk.take(64 * 16 * 32);
    e->pk_c3f = pk.take(32 * 9 * 64);
    e->pk_c3d = pk.take(32 * 9 * 64);
    e->pk_fc = pk.take((int64_t)H * std::max(e->fc_in, 4));
    for (int l = 0; l < d.rnn_layers; ++l) e->pk_whht.push_back(pk.take((int64_t)e->G_ * H * H));
    for (int l = 0; l < d.rnn_layers; ++l) e->pk_wiht.push_back(l == 0 ? -1 : pk.take((int64_t)e->G_ * H * H));
    e->pk_wih0 = pk.take((int64_t)e->G_ * H * e->rnn_ld);  // layer 0's W_ih with rows padded to rnn_ld (fused step projection of `act`)
    e->packed_floats = pk.used;

    Arena wk;
    const int64_t B = d.max_frames;
    const int64_t F = d.max_frames;  // worst case: every frame its own fragment
    const int64_t m1 = blind ? 0 : (int64_t)e->c1.Ho() * e->c1.Wo() * 32, m2 = blind ? 0 : (int64_t)e->c2.Ho() * e->c2.Wo() * 64, m3 = e->fc_in;
    e->w_a1 = wk.take(B * m1); e->w_a2 = wk.take(B * m2); e->w_a3 = wk.take(B * m3);
    e->w_da1 = wk.take(B * m1); e->w_da2 = wk.take(B * m2); e->w_da3 = wk.take(B * m3);
    e->w_rnnin = wk.take(B * e->rnn_ld); e->w_drnnin = wk.take(B * e->rnn_ld);
    e->w_hinit = wk.take((int64_t)d.rnn_layers * F * H); e->w_cinit = wk.take((int64_t)d.rnn_layers * F * H);
    for (int l = 0; l < d.rnn_layers; ++l) {
        e->w_gi.push_back(wk.take(B * e->G_ * H)); e->w_gates.push_back(wk.take(B * e->G_ * H));
        e->w_hn.push_back(wk.take(B * H)); e->w_hprev.push_back(wk.take(B * H));
        e->w_cprev.push_back(wk.take(B * H)); e->w_c.push_back(wk.take(B * H)); e->w_out.push_back(wk.take(B * H));
        e->w_dgi.push_back(wk.take(B * e->G_ * H)); e->w_dgh.push_back(wk.take(B * e->G_ * H));
        e->w_dlayer.push_back(wk.take(B * H));
    }
    e->w_probs = wk.take(B * 8); e->w_logitsn = wk.take(B * 8); e->w_dzv = wk.take(B * 8); e->w_dv = wk.take(B);
    e->w_dfeat = wk.take(B * H); e->w_scratch = wk.take((int64_t)std::max(3, 2 * d.rnn_layers) * F * H);
    e->w_value = wk.take(B); e->w_logp = wk.take(B); e->w_ent = wk.take(B);
    e->w_hmask = wk.take((int64_t)2 * d.rnn_layers * d.max_envs * H);
    e->w_gistep = wk.take((int64_t)d.max_envs * e->G_ * H);
    e->w_step_h = wk.take((int64_t)d.max_envs * H * 2);
    // split-K / column-sum scratch: big enough for 64 slabs of the largest weight-gradient (fc) or 1024 colsum rows
    e->ws_floats = std::max<int64_t>((int64_t)16 << 20, (int64_t)8 * H * e->fc_in / 4);
    e->w_ws = wk.take(e->ws_floats);
    // second stream of the time-major chunked recurrence: its own split-K scratch, the dense per-frame episode-start mask, an iota
    e->ws2_floats = e->ws_floats;  // same cap as the first stream's: the split-K plans (hence the bits) must not depend on the stream
    e->w_ws2 = wk.take(e->ws2_floats);
    e->w_fmask = wk.take((B + 3) / 4 + 64);
    e->w_iota = wk.take(B + 64);
    e->work_floats = wk.used;
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int hab_policy_create(const hab_policy_desc* desc, hab_policy** out) {
    if (!desc || !out) return HAB_ERR_ARG;
    if (desc->hidden <= 0 || desc->hidden % 64 || desc->num_actions <= 0 || desc->num_actions > 8 || desc->max_frames <= 0 ||
        desc->max_envs <= 0 || desc->rnn_layers <= 0 || desc->goal_dim < 0 ||
        ((desc->has_rgb || desc->has_depth || desc->has_semantic) && (desc->H <= 0 || desc->W <= 0)))
        return HAB_ERR_ARG;
    if (desc->action_dist != HAB_DIST_CATEGORICAL && (desc->action_dist != HAB_DIST_GAUSSIAN || desc->arch != HAB_ARCH_RESNET))
        return HAB_ERR_UNSUPPORTED;  // PointNavBaselinePolicy never builds a Gaussian head (rl/ppo/policy.py:439-460)
    // the Gaussian head kernels (heads.hip: saved[f][16], dz stride 8, std column sums at dz + A) hold 2A <= 8 values per frame
    if (desc->action_dist == HAB_DIST_GAUSSIAN && desc->num_actions > 4) return HAB_ERR_UNSUPPORTED;
    hab_policy* e = new hab_policy();
    e->d = *desc;
    int rc = HAB_ERR_UNSUPPORTED;
    if (desc->arch == HAB_ARCH_SIMPLE_CNN) rc = build_baseline(e);
    else if (desc->arch == HAB_ARCH_RESNET) rc = build_resnet(e);
    if (rc != HAB_OK) { destroy_resnet(e); delete e; return rc; }
    *out = e;
    return HAB_OK;
}
extern "C" void hab_policy_destroy(hab_policy* e) {
    if (!e) return;
    for (auto& ev : e->probe_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : e->evs) (void)hipEventDestroy(ev);
    if (e->s2) (void)hipStreamDestroy(e->s2);
    destroy_resnet(e);
    delete e;
}
extern "C" int hab_policy_num_params(const hab_policy* e) { return e ? (int)e->params.size() : HAB_ERR_ARG; }
extern "C" int hab_policy_param_info(const hab_policy* e, int i, char* name, int name_cap, int64_t* shape4, int* ndim,
                                     int64_t* offset_floats) {
    if (!e || i < 0 || i >= (int)e->params.size() || !name || name_cap <= 0) return HAB_ERR_ARG;
    const ParamSpec& s = e->params[i];
    strncpy(name, s.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = s.shape[k];
    if (ndim) *ndim = s.ndim;
    if (offset_floats) *offset_floats = s.offset;
    return HAB_OK;
}
extern "C" int hab_policy_param_is_buffer(const hab_policy* e, int i) {
    if (!e || i < 0 || i >= (int)e->params.size()) return HAB_ERR_ARG;
    return e->params[i].is_buffer;
}
extern "C" int hab_policy_set_training(hab_policy* e, int training) {
    if (!e) return HAB_ERR_ARG;
    e->training = training ? 1 : 0;
    return HAB_OK;
}
extern "C" int hab_policy_set_grad_ready(hab_policy* e, hab_grad_ready_fn fn, void* ctx) {
    if (!e) return HAB_ERR_ARG;
    e->grad_ready_cb = fn; e->grad_ready_ctx = ctx;
    return HAB_OK;
}
// Everything from parameter `first_param` to the end of the gradient arena has been written (enqueued) for this backward.
void grad_tail_ready(hab_policy* e, int first_param) {
    if (e->comm) {  // device-side exchange: this tail goes out on the communicator's stream now, beside the rest of backward
        // HAB_NO_GRAD_OVERLAP=1 (rl/ddppo/ddppo.py): nothing is sent early, hab_policy_grad_sync exchanges the whole arena in one message
        static const bool no_overlap = hab_env_flag("HAB_NO_GRAD_OVERLAP");
        if (no_overlap || e->comm_err != HAB_OK) return;
        const int64_t first = e->g(first_param) - e->G;
        const int64_t end = e->comm_first < 0 ? (int64_t)e->param_floats : e->comm_first;
        if (first >= end) return;
        const int rc = comm_exchange_async(e->comm, e->G, first, end - first, e->cur_stream);
        if (rc == HAB_OK) e->comm_first = first;
        else e->comm_err = rc;  // surfaced by hab_policy_backward / hab_policy_grad_sync
        return;
    }
    if (!e->grad_ready_cb) return;
    const int64_t first = e->g(first_param) - e->G;
    e->grad_ready_cb(first, (int64_t)e->param_floats - first, e->grad_ready_ctx);
}
extern "C" int hab_policy_set_allreduce(hab_policy* e, hab_allreduce_fn fn, void* ctx, int world_size) {
    if (!e || world_size < 1) return HAB_ERR_ARG;
    e->allreduce_cb = fn; e->allreduce_ctx = ctx; e->world_size = world_size;
    return HAB_OK;
}
extern "C" int64_t hab_policy_param_floats(const hab_policy* e) { return e ? e->param_floats : -1; }
extern "C" int64_t hab_policy_packed_floats(const hab_policy* e) { return e ? e->packed_floats : -1; }
extern "C" int64_t hab_policy_work_floats(const hab_policy* e) { return e ? e->work_floats : -1; }

extern "C" int hab_policy_bind(hab_policy* e, float* params, float* grads, float* packed, float* work, int64_t work_floats) {
    if (!e || !params || !packed || !work) return HAB_ERR_ARG;
    if (work_floats < e->work_floats) return HAB_ERR_ARG;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)packed | (uintptr_t)work) & 255) != 0) return HAB_ERR_ARG;
    e->P = params; e->G = grads; e->PK = packed; e->WK = work; e->work_bound = work_floats;
    return HAB_OK;
}

// Re-derive the kernel-layout weight copies from the master parameters (call after every change
// of the parameters: optimiser step, load_state_dict, broadcast).
extern "C" int hab_policy_repack(hab_policy* e, hipStream_t stream) {
    if (!e || !e->P) return HAB_ERR_ARG;
    if (e->rn) return resnet_repack(e, stream);
    const int H = e->d.hidden;
    if (e->pk_wih0 >= 0) HAB_TRY(pad_rows(e->p(e->i_wih[0]), e->PK + e->pk_wih0, e->G_ * H, e->rnn_in, e->rnn_ld, stream));
    if (e->Cin == 0) {  // blind baseline policy: only the recurrent weights have a kernel-layout copy
        for (int l = 0; l < e->L; ++l) HAB_TRY(transpose2d(e->p(e->i_whh[l]), e->PK + e->pk_whht[l], e->G_ * H, H, stream));
        for (int l = 1; l < e->L; ++l) HAB_TRY(transpose2d(e->p(e->i_wih[l]), e->PK + e->pk_wiht[l], e->G_ * H, H, stream));
        return HAB_OK;
    }
    HAB_TRY(repack_conv(e->p(e->i_c1w), e->PK + e->pk_c1f, nullptr, 32, e->Cin, 8, 8, e->Cin, stream));
    if (e->pk_c1img >= 0) {  // (was one launch in front of EVERY conv1 call: 12 -> 11 launches per rollout step)
        const int rc = obs_conv_weight_image(e->PK + e->pk_c1f, 32, 8, 8, e->Cin, e->PK + e->pk_c1img, stream);
        if (rc != HAB_OK) return rc == 1 ? HAB_ERR_UNSUPPORTED : rc;
    }
    HAB_TRY(repack_conv(e->p(e->i_c2w), e->PK + e->pk_c2f, e->PK + e->pk_c2d, 64, 32, 4, 4, 32, stream));
    HAB_TRY(repack_conv(e->p(e->i_c3w), e->PK + e->pk_c3f, e->PK + e->pk_c3d, 32, 64, 3, 3, 64, stream));
    HAB_TRY(repack_flatten(e->p(e->i_fcw), e->PK + e->pk_fc, H, 32, e->fc_in / 32, stream));
    for (int l = 0; l < e->L; ++l) HAB_TRY(transpose2d(e->p(e->i_whh[l]), e->PK + e->pk_whht[l], e->G_ * H, H, stream));
    for (int l = 1; l < e->L; ++l) HAB_TRY(transpose2d(e->p(e->i_wih[l]), e->PK + e->pk_wiht[l], e->G_ * H, H, stream));
    return HAB_OK;
}

// ---- probes: HIP-event timing of one tagged kernel call site (bench.py's roofline leg) ----
extern "C" int hab_policy_probe_enable(hab_policy* e, int tag) {
    if (!e || tag >= 64) return HAB_ERR_ARG;
    e->probe_mask = tag < 0 ? 0 : (uint64_t)1 << tag;
    e->probe_used = 0;
    for (int i = 0; i < 64; ++i) { e->probe_flops[i] = 0.0; e->probe_bytes[i] = 0.0; }
    return HAB_OK;
}
extern "C" int hab_policy_probe_enable_mask(hab_policy* e, uint64_t mask) {
    if (!e) return HAB_ERR_ARG;
    e->probe_mask = mask;
    e->probe_used = 0;
    for (int i = 0; i < 64; ++i) { e->probe_flops[i] = 0.0; e->probe_bytes[i] = 0.0; }
    return HAB_OK;
}
static int probe_sum(hab_policy* e, int tag, double* total_ms, int* count) {
    double t = 0;
    int n = 0;
    for (size_t i = 0; i < e->probe_used; ++i) {
        if (tag >= 0 && e->probe_events[i].tag != tag) continue;
        hipError_t err = hipEventSynchronize(e->probe_events[i].second);
        if (err != hipSuccess) return (int)err;
        float ms = 0;
        err = hipEventElapsedTime(&ms, e->probe_events[i].first, e->probe_events[i].second);
        if (err != hipSuccess) return (int)err;
        t += ms;
        ++n;
    }
    *total_ms = t;
    *count = n;
    return HAB_OK;
}
extern "C" int hab_policy_probe_work(hab_policy* e, int tag, double* flops, double* bytes) {
    if (!e || tag < 0 || tag >= 64 || !flops || !bytes) return HAB_ERR_ARG;
    *flops = e->probe_flops[tag]; *bytes = e->probe_bytes[tag];
    return HAB_OK;
}
extern "C" int hab_policy_probe_read(hab_policy* e, double* total_ms, int* count) {
    if (!e || !total_ms || !count) return HAB_ERR_ARG;
    HAB_TRY(probe_sum(e, -1, total_ms, count));
    e->probe_used = 0;
    return HAB_OK;
}
extern "C" int hab_policy_probe_read_tag(hab_policy* e, int tag, double* total_ms, int* count) {
    if (!e || !total_ms || !count || tag < 0 || tag >= 64) return HAB_ERR_ARG;
    return probe_sum(e, tag, total_ms, count);
}


// ------------------------------------------------------------------------------------------
// Encoder forward on B frames (shared by act / evaluate): obs -> rnn_in[B][rnn_ld]
// ------------------------------------------------------------------------------------------
// f0 / nB: frames [f0, f0 + nB) of the minibatch only (the time-major chunked evaluate); rows may be null (dense frames: act)
static int encoder_forward(hab_policy* e, const hab_obs* obs, const uint8_t* masks, const int* rows, int B, hipStream_t s, int f0 = 0, int nB = -1) {
    if (e->rn) { Probe pr(e, HAB_PROBE_ENC_FWD, s); return resnet_encoder_forward(e, obs, masks, rows, B, s, f0, nB); }
    if (nB < 0) nB = B;
    float* W = e->WK;
    float* ws = W + e->w_ws;
    const int H = e->d.hidden;
    if (e->Cin == 0) {  // blind: rnn_in = the goal vector alone (policy.py:572-582)
        if (!rows && f0) return HAB_ERR_ARG;
        return gather_cols(obs->goal, e->d.goal_dim, rows ? rows + f0 : nullptr, W + e->w_rnnin + (int64_t)f0 * e->rnn_ld, e->rnn_ld, 0, e->d.goal_dim,
                           e->rnn_ld - e->rnn_in, nB, s);
    }
    const int64_t m1 = (int64_t)e->c1.Ho() * e->c1.Wo() * 32, m2 = (int64_t)e->c2.Ho() * e->c2.Wo() * 64, m3 = e->fc_in;
    float *a1 = W + e->w_a1 + f0 * m1, *a2 = W + e->w_a2 + f0 * m2, *a3 = W + e->w_a3 + f0 * m3, *rin = W + e->w_rnnin + (int64_t)f0 * e->rnn_ld;
    const int* rws = rows ? rows + f0 : nullptr;
    if (!rows && f0) return HAB_ERR_ARG;
    ObsView ov;
    ov.rgb = e->d.has_rgb ? obs->rgb : nullptr; ov.depth = e->d.has_depth ? obs->depth : nullptr; ov.rows = rws;
    ov.H = e->d.H; ov.W = e->d.W; ov.C = e->Cin;
    ConvDesc c1 = e->c1, c2 = e->c2, c3 = e->c3;
    c1.B = c2.B = c3.B = nB;
    { Probe pr(e, HAB_PROBE_CONV1_FWD, s);
      HAB_TRY(obs_conv_fwd(c1, ov, e->PK + e->pk_c1f, e->p(e->i_c1b), a1, 1, ws, e->ws_floats, s, e->pk_c1img >= 0 ? e->PK + e->pk_c1img : nullptr)); }
    { Probe pr(e, HAB_PROBE_CONV2_FWD, s);
      HAB_TRY(conv_fwd(c2, a1, e->PK + e->pk_c2f, e->p(e->i_c2b), a2, 1, ws, e->ws_floats, s)); }
    { Probe pr(e, HAB_PROBE_CONV3_FWD, s);
      HAB_TRY(conv_fwd(c3, a2, e->PK + e->pk_c3f, e->p(e->i_c3b), a3, 0, ws, e->ws_floats, s)); }
    { Probe pr(e, HAB_PROBE_FC_FWD, s);
      HAB_TRY(linear_fwd(a3, e->fc_in, e->PK + e->pk_fc, e->fc_in, e->p(e->i_fcb), rin, e->rnn_ld, nB, H, e->fc_in, 1, 0, ws, e->ws_floats, s)); }
    if (e->d.goal_dim > 0)
        HAB_TRY(gather_cols(obs->goal, e->d.goal_dim, rws, rin, e->rnn_ld, H, e->d.goal_dim, e->rnn_ld - e->rnn_in, nB, s));
    return HAB_OK;
}

// ---- time-major chunked recurrence: second stream + events ----
int tm_chunks_cfg() { static const int v = hab_env_int("HAB_RNN_CHUNKS", 4); return v; }
// ResNet policies: bit-identical to the packed form at one chunk (tests/test_gpu_determinism.py).  With one launch per step the time-major
// form measured SLOWER than the packed form on C3 (round 4: 28.5 k vs 29.1 k env-steps/s: all T = 128 steps of 32 rows against max_len ~ 107
// packed steps, and every chunk is another pass through ~60 encoder kernels).  With the persistent recurrence (rnn_persist.h, round 6) a
// chunk's steps are one launch per layer on the second stream and the balance tips for large minibatches -- same box, env-steps/s:
//   C3 (ResNet18, 4096 frames per minibatch): packed 32.27 k / 32.54 k, 1 chunk 32.68 k, 2 chunks 33.15 k / 33.24 k, 3 chunks 32.27 k, 4 chunks 32.87 k
//   C5 (ResNet50, 1024 frames per minibatch): packed 4.995 k, 1 chunk 5.003 k, 2 chunks 4.92 k, 3 chunks 4.74 k
// Default (HAB_RNN_CHUNKS_RESNET unset or < 0): 2 chunks from 4096 frames per minibatch on, packed below; k >= 0 forces k chunks
// (0 = packed).  HAB_RNN_CHUNKS = 0 / 1 still selects the packed form / one chunk for both policies.
int tm_chunks_resnet_cfg(int frames) {
    static const int v = hab_env_int("HAB_RNN_CHUNKS_RESNET", -1);
    const int base = tm_chunks_cfg();
    if (base <= 1) return base;
    if (v >= 0) return v;
    return frames >= 4096 ? 2 : 0;
}
static int tm_setup(hab_policy* e, int nev) {
    if (!e->s2) {
        // highest priority: the recurrence is a chain of ~7 us launches; each must be dispatched ahead of the queued waves of the large
        // contraction running beside it, or the chain inherits that kernel's tail
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipError_t err = hipStreamCreateWithPriority(&e->s2, hipStreamNonBlocking, hi);
        if (err != hipSuccess) return (int)err;
    }
    while ((int)e->evs.size() < nev) {
        hipEvent_t ev;
        hipError_t err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (err != hipSuccess) return (int)err;
        e->evs.push_back(ev);
    }
    return HAB_OK;
}
// everything enqueued on `from` so far happens before whatever is enqueued on `to` from now on
static int tm_order(hab_policy* e, int ev, hipStream_t from, hipStream_t to) {
    hipError_t err = hipEventRecord(e->evs[ev], from);
    if (err != hipSuccess) return (int)err;
    err = hipStreamWaitEvent(to, e->evs[ev], 0);
    return err == hipSuccess ? HAB_OK : (int)err;
}

static RnnLayerParams layer_params(hab_policy* e, int l) {
    RnnLayerParams lp;
    lp.w_ih = e->p(e->i_wih[l]); lp.w_hh = e->p(e->i_whh[l]); lp.b_ih = e->p(e->i_bih[l]); lp.b_hh = e->p(e->i_bhh[l]);
    lp.w_hh_t = e->PK + e->pk_whht[l];
    if (e->G) { lp.dw_ih = e->g(e->i_wih[l]); lp.dw_hh = e->g(e->i_whh[l]); lp.db_ih = e->g(e->i_bih[l]); lp.db_hh = e->g(e->i_bhh[l]); }
    else { lp.dw_ih = lp.dw_hh = lp.db_ih = lp.db_hh = nullptr; }
    lp.in_dim = l == 0 ? e->rnn_in : e->d.hidden;
    if (l == 0 && e->pk_wih0 >= 0) { lp.w_ih_pad = e->PK + e->pk_wih0; lp.w_ih_ld = e->rnn_ld; }
    else if (l > 0 && (e->d.hidden & 15) == 0) { lp.w_ih_pad = lp.w_ih; lp.w_ih_ld = e->d.hidden; }
    return lp;
}
static RnnWork layer_work(hab_policy* e, int l) {
    float* W = e->WK;
    RnnWork wk;
    wk.gi = W + e->w_gi[l]; wk.gates = W + e->w_gates[l]; wk.hn = W + e->w_hn[l]; wk.hprev = W + e->w_hprev[l];
    wk.cprev = W + e->w_cprev[l]; wk.c = W + e->w_c[l]; wk.out = W + e->w_out[l]; wk.dgi = W + e->w_dgi[l];
    wk.dgh = W + e->w_dgh[l];
    return wk;
}

extern "C" int hab_policy_encode(hab_policy* e, const hab_obs* obs, int n, float* out, hipStream_t stream) {
    if (!e || !e->P || !obs || !out || n <= 0 || n > e->d.max_frames) return HAB_ERR_ARG;
    if (!e->rn) return HAB_ERR_UNSUPPORTED;  // the baseline net has no visual_features input (rl/ppo/policy.py:557-589)
    Probe pr(e, HAB_PROBE_ENC_FWD, stream);
    e->save_acts = 0;
    const int rc = resnet_encode(e, obs, n, out, stream);
    e->save_acts = 1;
    return rc;
}
extern "C" int hab_policy_visual_feature_shape(const hab_policy* e, int* c, int* hf, int* wf) {
    if (!e || !c || !hf || !wf) return HAB_ERR_ARG;
    if (!e->rn) return HAB_ERR_UNSUPPORTED;
    return resnet_feature_shape(e, c, hf, wf);
}

// ------------------------------------------------------------------------------------------
// NetPolicy.act / get_value (rl/ppo/policy.py:324-359) on n envs, all tensors dense over envs.
// hidden_in/out: (n, Lh, H) with Lh = layers (GRU) or 2*layers (LSTM: h layers then c layers).
// ------------------------------------------------------------------------------------------
extern "C" int hab_policy_act(hab_policy* e, const hab_obs* obs, const float* hidden_in, const uint8_t* masks,
                              const float* exp_noise, int deterministic, int n, float* values, int64_t* actions,
                              float* action_log_probs, float* hidden_out, float* probs_out, hipStream_t stream) {
    if (!e || !e->P || !obs || !hidden_in || !masks || !values || n <= 0 || n > e->d.max_envs) return HAB_ERR_ARG;
    float* W = e->WK;
    const int H = e->d.hidden, L = e->L;
    const int Lh = e->d.rnn_type == HAB_RNN_LSTM ? 2 * L : L;
    e->save_acts = 0;  // no backward follows an act: fused kernels skip the copies kept for it
    const int enc_rc = encoder_forward(e, obs, masks, nullptr, n, stream);
    e->save_acts = 1;
    HAB_TRY(enc_rc);
    const float* x = W + e->w_rnnin;
    int ldx = e->rnn_ld;
    // The episode-start mask (h * masks, rnn_state_encoder.py:308-311) is applied to the state operand inside the step kernel and
    // every layer reads / writes the caller's (n, Lh, H) tensors in place through row strides: no masked / dense staging copies.
    if (hidden_out) {  // in-place update is not possible: a layer's workgroups read whole state rows while others write them
        const float* a0 = hidden_in; const float* a1 = hidden_in + (size_t)n * Lh * H;
        if (hidden_out < a1 && a0 < hidden_out + (size_t)n * Lh * H) return HAB_ERR_ARG;
    }
    float* step_h = W + e->w_step_h;
    for (int l = 0; l < L; ++l) {
        RnnLayerParams lp = layer_params(e, l);
        float* hout = hidden_out ? hidden_out + (size_t)l * H : step_h + (size_t)(l & 1) * n * H;
        const int hstride = hidden_out ? Lh * H : H;
        float* cout = (hidden_out && e->d.rnn_type == HAB_RNN_LSTM) ? hidden_out + (size_t)(L + l) * H : nullptr;
        HAB_TRY(rnn_step_layer_forward(e->d.rnn_type, H, lp, x, ldx, hidden_in + (size_t)l * H, Lh * H,
                                       e->d.rnn_type == HAB_RNN_LSTM ? hidden_in + (size_t)(L + l) * H : nullptr, Lh * H, masks, n,
                                       W + e->w_gistep, hout, hstride, cout, Lh * H, W + e->w_ws, e->ws_floats, stream));
        x = hout;  // next layer's input / the heads' features, read with the same row stride
        ldx = hstride;
    }
    if (e->d.action_dist == HAB_DIST_GAUSSIAN) {
        GaussHeadsArgs ga;
        ga.B = n; ga.H = H; ga.A = e->d.num_actions; ga.K = e->head_K; ga.flags = e->d.gauss_flags;
        ga.min_std = e->d.gauss_min_std; ga.max_std = e->d.gauss_max_std;
        ga.mode = actions ? (deterministic ? 2 : 1) : 2;
        ga.feats = x; ga.feats_ld = ldx; ga.w = e->p(e->i_aw); ga.b = e->p(e->i_ab);
        ga.std_param = e->i_astd >= 0 ? e->p(e->i_astd) : nullptr; ga.w_critic = e->p(e->i_cw); ga.b_critic = e->p(e->i_cb);
        ga.actions_in = nullptr; ga.rows = nullptr; ga.noise = exp_noise;
        ga.actions_out = actions ? reinterpret_cast<float*>(actions) : W + e->w_dzv;
        ga.value = values; ga.logp = action_log_probs ? action_log_probs : W + e->w_logp; ga.entropy = nullptr; ga.saved = nullptr;
        if (ga.mode == 1 && !exp_noise) return HAB_ERR_ARG;
        if (probs_out) return HAB_ERR_ARG;
        return gauss_heads_forward(ga, stream);
    }
    HeadsArgs ha;
    ha.B = n; ha.H = H; ha.A = e->d.num_actions;
    ha.mode = actions ? (deterministic ? 2 : 1) : 2;
    ha.feats = x; ha.feats_ld = ldx; ha.w_actor = e->p(e->i_aw); ha.b_actor = e->p(e->i_ab); ha.w_critic = e->p(e->i_cw); ha.b_critic = e->p(e->i_cb);
    ha.actions_in = nullptr; ha.rows = nullptr; ha.noise = exp_noise;
    ha.actions_out = actions ? actions : reinterpret_cast<int64_t*>(W + e->w_dzv);
    ha.value = values; ha.logp = action_log_probs ? action_log_probs : W + e->w_logp; ha.entropy = nullptr;
    ha.probs = probs_out ? W + e->w_probs : nullptr; ha.logits_n = W + e->w_logitsn;
    if (ha.mode == 1 && !exp_noise) return HAB_ERR_ARG;
    HAB_TRY(heads_forward(ha, stream));
    if (probs_out) HAB_TRY(copy_rows(W + e->w_probs, nullptr, 8, probs_out, 8, n, 8, stream));
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------
// NetPolicy.evaluate_actions (rl/ppo/policy.py:361-402) on B = T*n frames gathered from the
// rollout arena through rows[f]; activations are kept for hab_policy_backward.
// ------------------------------------------------------------------------------------------
extern "C" int hab_policy_evaluate(hab_policy* e, const hab_obs* obs, const int* rows, const float* hidden0,
                                   int hidden_env_stride, const uint8_t* masks, const int64_t* actions,
                                   const hab_pack_info* pack, int B, int n, float* value, float* log_prob, float* entropy,
                                   hipStream_t stream) {
    if (!e || !e->P || !obs || !hidden0 || !masks || !actions || !pack || B <= 0 || n <= 0 || B > e->d.max_frames) return HAB_ERR_ARG;
    if (!pack->env_first_frame && (B % n)) return HAB_ERR_ARG;  // time-major T x n minibatch unless the pack info names the env frames
    if (pack->P != B || pack->F <= 0 || pack->F > B || pack->max_len <= 0) return HAB_ERR_ARG;
    float* W = e->WK;
    const int H = e->d.hidden, L = e->L;
    // Time-major chunked form (rnn.hip): a regular T x n minibatch of the SimpleCNN policy is cut into time chunks; the encoder of
    // chunk c + 1 runs on `stream` while the recurrence walks chunk c on the engine's second stream.  Bit-identical to the packed form.
    const int T = B / n;
    // (ResNet policies too since round 4: their encoder runs per chunk behind a whole-batch ingest, engine_resnet.hip)
    int NC = (e->Cin > 0 && rows && !pack->env_first_frame && (B % n) == 0 && T >= 2 && e->w_ws2 >= 0 && (int64_t)L * 2 * n <= 3 * (int64_t)e->d.max_frames)
                 ? (e->rn ? tm_chunks_resnet_cfg(B) : tm_chunks_cfg()) : 0;
    if (NC > T) NC = T;
    const float* x = W + e->w_rnnin;
    int ldx = e->rnn_ld;
    if (NC > 0) {
        HAB_TRY(tm_setup(e, 2 * NC + 4));
        hipStream_t sB = e->s2;
        uint8_t* fmask = reinterpret_cast<uint8_t*>(W + e->w_fmask);
        int* iota = reinterpret_cast<int*>(W + e->w_iota);
        HAB_TRY(rnn_tm_prepare(masks, rows, B, fmask, iota, stream));
        for (int l = 0; l < L; ++l) {  // state entering t = 0: env j of the minibatch is frame j; its arena row is rows[j]
            HAB_TRY(rnn_frag_init(hidden0 + (size_t)l * H, rows, hidden_env_stride, masks, rows, nullptr, nullptr, n, H,
                                  W + e->w_hinit + (size_t)l * n * H, stream, nullptr));
            if (e->d.rnn_type == HAB_RNN_LSTM)
                HAB_TRY(rnn_frag_init(hidden0 + (size_t)(L + l) * H, rows, hidden_env_stride, masks, rows, nullptr, nullptr, n, H,
                                      W + e->w_cinit + (size_t)l * n * H, stream, nullptr));
        }
        const int Tc = (T + NC - 1) / NC;
        for (int c = 0; c < NC; ++c) {
            const int t0 = c * Tc, t1 = std::min(T, t0 + Tc);
            if (t0 >= t1) break;
            HAB_TRY(encoder_forward(e, obs, masks, rows, B, stream, t0 * n, (t1 - t0) * n));
            HAB_TRY(tm_order(e, c, stream, sB));  // also orders everything before this evaluate (parameters, pack info) ahead of stream B
            const float* xl = W + e->w_rnnin;
            int ldl = e->rnn_ld;
            for (int l = 0; l < L; ++l) {
                RnnLayerParams lp = layer_params(e, l);
                RnnWork wk = layer_work(e, l);
                Probe pr(e, HAB_PROBE_RNN_FWD, sB);
                HAB_TRY(rnn_tm_layer_forward(e->d.rnn_type, H, lp, wk, xl, ldl, W + e->w_hinit + (size_t)l * n * H,
                                             W + e->w_cinit + (size_t)l * n * H, fmask, n, T, t0, t1, W + e->w_ws2, e->ws2_floats, sB));
                xl = wk.out;
                ldl = H;
            }
        }
        HAB_TRY(tm_order(e, NC, sB, stream));  // the heads (and everything after this evaluate) wait for the recurrence
        x = W + e->w_out[L - 1];
        ldx = H;
    } else {
    HAB_TRY(encoder_forward(e, obs, masks, rows, B, stream));
    PackInfo pk;
    pk.select_inds = pack->select_inds; pk.step_offsets = pack->step_offsets_host; pk.num_seqs_at_step = pack->num_seqs_at_step_host;
    pk.frag_env = pack->frag_env; pk.frag_start = pack->frag_start; pk.P = pack->P; pk.F = pack->F; pk.max_len = pack->max_len;
    pk.n_envs = n;
    for (int l = 0; l < L; ++l) {
        float* hinit = W + e->w_hinit + (size_t)l * pk.F * H;
        float* cinit = W + e->w_cinit + (size_t)l * pk.F * H;
        // env j of the minibatch is frame j (t = 0); its arena row is rows[j]
        HAB_TRY(rnn_frag_init(hidden0 + (size_t)l * H, rows, hidden_env_stride, masks, rows, pk.frag_env, pk.frag_start, pk.F, H,
                              hinit, stream, pack->env_first_frame));
        if (e->d.rnn_type == HAB_RNN_LSTM)
            HAB_TRY(rnn_frag_init(hidden0 + (size_t)(L + l) * H, rows, hidden_env_stride, masks, rows, pk.frag_env,
                                  pk.frag_start, pk.F, H, cinit, stream, pack->env_first_frame));
    }
    // several layers: one launch per packed step for ALL layers (layer l one step behind layer l - 1), rnn.hip
    static const int wave_cfg = hab_env_int("HAB_RNN_WAVE", 1);
    e->last_wave = false;
    if (wave_cfg && L >= 2 && L <= 4) {
        RnnLayerParams lps[4]; RnnWork wks[4]; const float* hin[4]; const float* cin[4];
        for (int l = 0; l < L; ++l) {
            lps[l] = layer_params(e, l); wks[l] = layer_work(e, l);
            hin[l] = W + e->w_hinit + (size_t)l * pk.F * H; cin[l] = W + e->w_cinit + (size_t)l * pk.F * H;
        }
        Probe pr(e, HAB_PROBE_RNN_FWD, stream);
        const int rcw = rnn_seq_wave_forward(e->d.rnn_type, H, L, lps, wks, x, ldx, hin, cin, pk, W + e->w_ws, e->ws_floats, stream);
        if (rcw != 0 && rcw != 1) return rcw;
        e->last_wave = rcw == 0;
        if (e->last_wave) { x = wks[L - 1].out; ldx = H; }
    }
    if (!e->last_wave)
    for (int l = 0; l < L; ++l) {
        float* hinit = W + e->w_hinit + (size_t)l * pk.F * H;
        float* cinit = W + e->w_cinit + (size_t)l * pk.F * H;
        RnnLayerParams lp = layer_params(e, l);
        RnnWork wk = layer_work(e, l);
        Probe pr(e, HAB_PROBE_RNN_FWD, stream);
        HAB_TRY(rnn_seq_layer_forward(e->d.rnn_type, H, lp, wk, x, ldx, hinit, cinit, pk, W + e->w_ws, e->ws_floats, stream));
        x = wk.out;
        ldx = H;
    }
    }
    e->last_tm = NC;
    if (e->d.action_dist == HAB_DIST_GAUSSIAN) {
        GaussHeadsArgs ga;
        ga.B = B; ga.H = H; ga.A = e->d.num_actions; ga.K = e->head_K; ga.flags = e->d.gauss_flags; ga.mode = 0;
        ga.min_std = e->d.gauss_min_std; ga.max_std = e->d.gauss_max_std;
        ga.feats = x; ga.feats_ld = ldx; ga.w = e->p(e->i_aw); ga.b = e->p(e->i_ab);
        ga.std_param = e->i_astd >= 0 ? e->p(e->i_astd) : nullptr; ga.w_critic = e->p(e->i_cw); ga.b_critic = e->p(e->i_cb);
        ga.actions_in = reinterpret_cast<const float*>(actions); ga.rows = rows; ga.noise = nullptr; ga.actions_out = nullptr;
        ga.value = value ? value : W + e->w_value; ga.logp = log_prob ? log_prob : W + e->w_logp;
        ga.entropy = entropy ? entropy : W + e->w_ent; ga.saved = W + e->w_gsaved;
        HAB_TRY(gauss_heads_forward(ga, stream));
    } else {
    HeadsArgs ha;
        ha.B = B; ha.H = H; ha.A = e->d.num_actions; ha.mode = 0;
        ha.feats = x; ha.feats_ld = ldx; ha.w_actor = e->p(e->i_aw); ha.b_actor = e->p(e->i_ab); ha.w_critic = e->p(e->i_cw); ha.b_critic = e->p(e->i_cb);
        ha.actions_in = actions; ha.rows = rows; ha.noise = nullptr; ha.actions_out = nullptr;
        ha.value = value ? value : W + e->w_value; ha.logp = log_prob ? log_prob : W + e->w_logp;
        ha.entropy = entropy ? entropy : W + e->w_ent;
        ha.probs = W + e->w_probs; ha.logits_n = W + e->w_logitsn;
        HAB_TRY(heads_forward(ha, stream));
    }
    e->last_B = B;
    e->last_n = pack->env_first_frame ? 0 : n;  // (no "last n frames = final step of every env" in a VER minibatch)
    e->last_masks = masks;
    return HAB_OK;
}

// Final hidden state of the last evaluate: (n, Lh, H) -- rnn_state_encoder.py:262-275.
extern "C" int hab_policy_final_hidden(hab_policy* e, float* hidden_out, hipStream_t stream) {
    if (!e || !hidden_out || e->last_B <= 0 || e->last_n <= 0) return HAB_ERR_ARG;
    const int H = e->d.hidden, L = e->L, n = e->last_n, B = e->last_B;
    const int Lh = e->d.rnn_type == HAB_RNN_LSTM ? 2 * L : L;
    for (int l = 0; l < L; ++l) {
        HAB_TRY(copy_rows(e->WK + e->w_out[l] + (size_t)(B - n) * H, nullptr, H, hidden_out + (size_t)l * H, Lh * H, n, H, stream));
        if (e->d.rnn_type == HAB_RNN_LSTM)
            HAB_TRY(copy_rows(e->WK + e->w_c[l] + (size_t)(B - n) * H, nullptr, H, hidden_out + (size_t)(L + l) * H, Lh * H, n, H, stream));
    }
    return HAB_OK;
}

// ------------------------------------------------------------------------------------------
// Backward of the last hab_policy_evaluate: given dL/dvalue, dL/dlog_prob, dL/dentropy per frame,
// writes EVERY parameter gradient into the gradient arena (overwrites; no accumulation).
// ------------------------------------------------------------------------------------------
// dst[i] += src[i]  /  dst[f][c] += src[f][c] * (x[f][c] > 0) for c < H (the ReLU of the visual fc sits between perception_embed and x)
__global__ void add_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] += src[i];
}
__global__ void add_masked_cols_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, const float* __restrict__ x, int ldx,
                                       int B, int H) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i / H), c = (int)(i % H);
        if (x[(size_t)f * ldx + c] > 0.f) dst[(size_t)f * ldd + c] += src[i];
    }
}
// Auxiliary-loss hook (rl/ppo/policy.py:253-291,386-394; rl/ppo/ppo.py:248): d_rnn_output / d_perception_embed are [B][H] device
// arrays in the frame order of the last evaluate (either may be null).  They are added where those tensors sit in the backward chain --
// behind the heads' gradient wrt the features, and (through the visual fc's ReLU) onto the gradient wrt the recurrent encoder's input --
// by the NEXT hab_policy_backward, which forgets them.  Packed form only (the autograd bridge evaluates dense batches).
extern "C" int hab_policy_set_extra_grads(hab_policy* e, const float* d_rnn_output, const float* d_perception_embed) {
    if (!e) return HAB_ERR_ARG;
    if (d_perception_embed && e->Cin == 0) return HAB_ERR_ARG;  // a blind net has no perception embedding
    e->xg_feat = d_rnn_output; e->xg_perc = d_perception_embed;
    return HAB_OK;
}

static int policy_backward_impl(hab_policy* e, const hab_obs* obs, const int* rows, const int64_t* actions,
                                const hab_pack_info* pack, const float* d_value, const float* d_log_prob,
                                const float* d_entropy, hipStream_t stream) {
    if (!e || !e->P || !e->G || !obs || !actions || !pack || !d_value || !d_log_prob || !d_entropy || e->last_B <= 0)
        return HAB_ERR_ARG;
    e->comm_err = HAB_OK;
    float* W = e->WK;
    float* ws = W + e->w_ws;
    const int H = e->d.hidden, L = e->L, B = e->last_B, A = e->d.num_actions;
    e->cur_stream = stream;
    e->comm_first = -1;
    PackInfo pk;
    pk.select_inds = pack->select_inds; pk.step_offsets = pack->step_offsets_host; pk.num_seqs_at_step = pack->num_seqs_at_step_host;
    pk.frag_env = pack->frag_env; pk.frag_start = pack->frag_start; pk.P = pack->P; pk.F = pack->F; pk.max_len = pack->max_len;
    pk.n_envs = e->last_n;
    const float* feats = W + e->w_out[L - 1];
    if (e->d.action_dist == HAB_DIST_GAUSSIAN) {
        const int K = e->head_K;
        GaussHeadsBwdArgs gb;
        gb.B = B; gb.H = H; gb.A = A; gb.K = K; gb.d_value = d_value; gb.d_logp = d_log_prob; gb.d_entropy = d_entropy;
        gb.actions = reinterpret_cast<const float*>(actions); gb.rows = rows; gb.saved = W + e->w_gsaved;
        gb.w = e->p(e->i_aw); gb.w_critic = e->p(e->i_cw); gb.dfeat = W + e->w_dfeat; gb.dz = W + e->w_dzv; gb.dv_out = W + e->w_dv;
        HAB_TRY(gauss_heads_backward(gb, stream));
        HAB_TRY(linear_wgrad(W + e->w_dzv, 8, feats, H, e->g(e->i_aw), H, B, K, H, 0, 0, 0, ws, e->ws_floats, stream));
        HAB_TRY(colsum(W + e->w_dzv, 8, B, K, e->g(e->i_ab), 0, ws, e->ws_floats, stream));
        if (e->i_astd >= 0) HAB_TRY(colsum(W + e->w_dzv + A, 8, B, A, e->g(e->i_astd), 0, ws, e->ws_floats, stream));
        HAB_TRY(linear_wgrad(W + e->w_dv, 1, feats, H, e->g(e->i_cw), H, B, 1, H, 0, 0, 0, ws, e->ws_floats, stream));
        HAB_TRY(colsum(W + e->w_dv, 1, B, 1, e->g(e->i_cb), 0, ws, e->ws_floats, stream));
    } else {
    HeadsBwdArgs hb;
        hb.B = B; hb.H = H; hb.A = A; hb.d_value = d_value; hb.d_logp = d_log_prob; hb.d_entropy = d_entropy;
        hb.actions = actions; hb.rows = rows; hb.probs = W + e->w_probs; hb.logits_n = W + e->w_logitsn;
        hb.w_actor = e->p(e->i_aw); hb.w_critic = e->p(e->i_cw); hb.dfeat = W + e->w_dfeat; hb.dzv = W + e->w_dzv; hb.dv_out = W + e->w_dv;
        HAB_TRY(heads_backward(hb, stream));
        HAB_TRY(linear_wgrad(W + e->w_dzv, 8, feats, H, e->g(e->i_aw), H, B, A, H, 0, 0, 0, ws, e->ws_floats, stream));
        HAB_TRY(linear_wgrad(W + e->w_dv, 1, feats, H, e->g(e->i_cw), H, B, 1, H, 0, 0, 0, ws, e->ws_floats, stream));
        HAB_TRY(colsum(W + e->w_dzv, 8, B, A, e->g(e->i_ab), 0, ws, e->ws_floats, stream));
        HAB_TRY(colsum(W + e->w_dv, 1, B, 1, e->g(e->i_cb), 0, ws, e->ws_floats, stream));
    }
    if (e->xg_feat || e->xg_perc) {
        if (e->last_tm > 0) return HAB_ERR_UNSUPPORTED;  // (time-major chunks consume d_rnnin chunk by chunk: the bridge never takes that form)
        if (e->xg_feat) {
            add_rows_kernel<<<(int)std::min<long long>(1024, cdivl((long long)B * H, 256)), 256, 0, stream>>>(W + e->w_dfeat, e->xg_feat, (long long)B * H);
            HAB_LAUNCH_CHECK();
        }
    }
    ConvDesc c1 = e->c1, c2 = e->c2, c3 = e->c3;
    c1.B = c2.B = c3.B = B;
    const float* dfc = W + e->w_drnnin;
    if (e->last_tm > 0 && e->rn) {
        // ResNet policy, time-major form: BPTT walks the chunks on the second stream, the recurrent weight gradients follow there, and
        // the encoder's backward (whose weight gradients reduce over ALL frames: one pass over the whole minibatch) starts once
        // d_rnnin is complete.  The forward is where this form pays for the ResNets: the recurrence of chunk c runs under the encoder
        // of chunk c + 1.
        const int n = e->last_n, T = B / n, NC = e->last_tm, Tc = (T + NC - 1) / NC;
        hipStream_t sB = e->s2;
        const uint8_t* fmask = reinterpret_cast<const uint8_t*>(W + e->w_fmask);
        const int* iota = reinterpret_cast<const int*>(W + e->w_iota);
        float* ws2 = W + e->w_ws2;
        HAB_TRY(tm_order(e, NC + 1, stream, sB));  // the head gradients are in place
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * Tc, t1 = std::min(T, t0 + Tc);
            if (t0 >= t1) continue;
            for (int l = L - 1; l >= 0; --l) {
                RnnLayerParams lp = layer_params(e, l);
                RnnWork wk = layer_work(e, l);
                const float* dout = l == L - 1 ? W + e->w_dfeat : W + e->w_dlayer[l + 1];
                const float* xin = l == 0 ? W + e->w_rnnin : W + e->w_out[l - 1];
                const int ldx = l == 0 ? e->rnn_ld : H;
                float* dx = l == 0 ? W + e->w_drnnin : W + e->w_dlayer[l];
                Probe pr(e, HAB_PROBE_RNN_BWD, sB);
                HAB_TRY(rnn_tm_layer_backward(e->d.rnn_type, H, lp, wk, dout, dx, ldx, l == 0 ? xin : nullptr, ldx, H, fmask, iota, n, T, t0, t1,
                                              W + e->w_scratch + (size_t)l * 2 * n * H, ws2, e->ws2_floats, sB));
            }
        }
        for (int l = L - 1; l >= 0; --l) {
            RnnLayerParams lp = layer_params(e, l);
            RnnWork wk = layer_work(e, l);
            HAB_TRY(rnn_tm_layer_param_grads(e->d.rnn_type, H, lp, wk, l == 0 ? W + e->w_rnnin : W + e->w_out[l - 1], l == 0 ? e->rnn_ld : H, B, ws2,
                                             e->ws2_floats, sB));
        }
        HAB_TRY(tm_order(e, 2 * NC + 3, sB, stream));  // d_rnnin and the recurrent gradients are final
        Probe pr(e, HAB_PROBE_ENC_BWD, stream);
        return resnet_encoder_backward(e, obs, e->last_masks, rows, B, stream);
    }
    if (e->last_tm > 0 && !e->rn) {
        // Time-major chunked backward: BPTT walks the chunks from the last to the first on the second stream; behind each chunk the
        // DATA-gradient chain of its frames (fc, conv3, conv2 -- per-frame work) runs on `stream`; the weight gradients, which reduce over
        // all frames, follow once at the end (the recurrent ones on the second stream, beside the encoder's).
        const int n = e->last_n, T = B / n, NC = e->last_tm, Tc = (T + NC - 1) / NC;
        hipStream_t sB = e->s2;
        const uint8_t* fmask = reinterpret_cast<const uint8_t*>(W + e->w_fmask);
        const int* iota = reinterpret_cast<const int*>(W + e->w_iota);
        float* ws2 = W + e->w_ws2;
        const int64_t m1 = (int64_t)e->c1.Ho() * e->c1.Wo() * 32, m2 = (int64_t)e->c2.Ho() * e->c2.Wo() * 64, m3 = e->fc_in;
        HAB_TRY(tm_order(e, NC + 1, stream, sB));  // the head gradients are in place
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * Tc, t1 = std::min(T, t0 + Tc);
            if (t0 >= t1) continue;
            const int64_t f0 = (int64_t)t0 * n;
            const int nB = (t1 - t0) * n;
            for (int l = L - 1; l >= 0; --l) {
                RnnLayerParams lp = layer_params(e, l);
                RnnWork wk = layer_work(e, l);
                const float* dout = l == L - 1 ? W + e->w_dfeat : W + e->w_dlayer[l + 1];
                const float* xin = l == 0 ? W + e->w_rnnin : W + e->w_out[l - 1];
                const int ldx = l == 0 ? e->rnn_ld : H;
                float* dx = l == 0 ? W + e->w_drnnin : W + e->w_dlayer[l];
                Probe pr(e, HAB_PROBE_RNN_BWD, sB);
                HAB_TRY(rnn_tm_layer_backward(e->d.rnn_type, H, lp, wk, dout, dx, ldx, l == 0 ? xin : nullptr, ldx, H, fmask, iota, n, T, t0, t1,
                                              W + e->w_scratch + (size_t)l * 2 * n * H, ws2, e->ws2_floats, sB));
            }
            HAB_TRY(tm_order(e, NC + 2 + c, sB, stream));  // d_rnnin of the chunk's frames is final
            ConvDesc k2 = c2, k3 = c3;
            k2.B = k3.B = nB;
            { Probe pr(e, HAB_PROBE_FC_DGRAD, stream);
              HAB_TRY(linear_dgrad(dfc + f0 * e->rnn_ld, e->rnn_ld, e->PK + e->pk_fc, e->fc_in, nullptr, 0, 0, W + e->w_da3 + f0 * m3, e->fc_in, nB,
                                   e->fc_in, H, 0, ws, e->ws_floats, stream)); }
            { Probe pr(e, HAB_PROBE_CONV3_DGRAD, stream);
              HAB_TRY(conv_dgrad(k3, W + e->w_da3 + f0 * m3, e->PK + e->pk_c3d, W + e->w_a2 + f0 * m2, nullptr, W + e->w_da2 + f0 * m2, ws,
                                 e->ws_floats, stream)); }
            { Probe pr(e, HAB_PROBE_CONV2_DGRAD, stream);
              HAB_TRY(conv_dgrad(k2, W + e->w_da2 + f0 * m2, e->PK + e->pk_c2d, W + e->w_a1 + f0 * m1, nullptr, W + e->w_da1 + f0 * m1, ws,
                                 e->ws_floats, stream)); }
        }
        for (int l = L - 1; l >= 0; --l) {  // recurrent weight gradients over all frames: second stream, beside the encoder's below
            RnnLayerParams lp = layer_params(e, l);
            RnnWork wk = layer_work(e, l);
            HAB_TRY(rnn_tm_layer_param_grads(e->d.rnn_type, H, lp, wk, l == 0 ? W + e->w_rnnin : W + e->w_out[l - 1], l == 0 ? e->rnn_ld : H, B, ws2,
                                             e->ws2_floats, sB));
        }
        { Probe pr(e, HAB_PROBE_FC_WGRAD, stream);
          HAB_TRY(linear_wgrad(dfc, e->rnn_ld, W + e->w_a3, e->fc_in, e->g(e->i_fcw), e->fc_in, B, H, e->fc_in, 32, e->fc_in / 32, 0,
                               ws, e->ws_floats, stream)); }
        HAB_TRY(colsum(dfc, e->rnn_ld, B, H, e->g(e->i_fcb), 0, ws, e->ws_floats, stream));
        HAB_TRY(tm_order(e, 2 * NC + 3, sB, stream));  // recurrent gradients final before the tail of the arena is announced
        grad_tail_ready(e, e->i_fcw);
    } else {
    // recurrent layers, top down
    const float* dout = W + e->w_dfeat;
    bool waved = false;
    if (e->last_wave) {  // the forward ran the layers as a wavefront (upper layers' wk.gi do not exist): mirror it
        RnnLayerParams lps[4]; RnnWork wks[4]; const float* wiht[4];
        for (int l = 0; l < L; ++l) { lps[l] = layer_params(e, l); wks[l] = layer_work(e, l); wiht[l] = l ? e->PK + e->pk_wiht[l] : nullptr; }
        // blind0: the baseline net's rnn_in is the goal vector alone -- no gradient is wanted for it; nofc: no ReLU(visual fc) columns
        // at the head of rnn_in, i.e. nothing to mask (both blind forms; the blind ResNet net still wants d rnn_in for its embeddings)
        const bool blind0 = !e->rn && e->Cin == 0, nofc = e->Cin == 0;
        const float* x0 = W + e->w_rnnin;
        Probe pr(e, HAB_PROBE_RNN_BWD, stream);
        const int rcw = rnn_seq_wave_backward(e->d.rnn_type, H, L, lps, wiht, wks, x0, e->rnn_ld, dout, blind0 ? nullptr : W + e->w_drnnin, e->rnn_ld,
                                              nofc ? nullptr : x0, e->rnn_ld, nofc ? 0 : H, pk, W + e->w_scratch, ws, e->ws_floats, stream);
        if (rcw != 0) return rcw == 1 ? HAB_ERR_UNSUPPORTED : rcw;
        waved = true;
    }
    for (int l = L - 1; l >= 0 && !waved; --l) {
        RnnLayerParams lp = layer_params(e, l);
        RnnWork wk = layer_work(e, l);
        const float* x = l == 0 ? W + e->w_rnnin : W + e->w_out[l - 1];
        const int ldx = l == 0 ? e->rnn_ld : H;
        float* dx = l == 0 ? W + e->w_drnnin : W + e->w_dlayer[l];
        const int lddx = l == 0 ? e->rnn_ld : H;
        // layer 0: the first H columns of rnn_in are ReLU(fc) -> mask them here (fused ReLU backward); a blind policy's rnn_in is the
        // goal vector alone: no mask, and no gradient is wanted for it
        const bool blind0 = !e->rn && e->Cin == 0 && l == 0, nofc = e->Cin == 0;
        Probe pr(e, HAB_PROBE_RNN_BWD, stream);
        HAB_TRY(rnn_seq_layer_backward(e->d.rnn_type, H, lp, wk, x, ldx, dout, blind0 ? nullptr : dx, lddx, (l == 0 && !nofc) ? x : nullptr, ldx,
                                       nofc ? 0 : H, pk, W + e->w_scratch, ws, e->ws_floats, stream));
        dout = dx;
    }
    if (e->xg_perc) {  // d rnn_in[:, :H] += d perception_embed, through the ReLU between them (the mask the layer-0 backward applied to its own part)
        add_masked_cols_kernel<<<(int)std::min<long long>(1024, cdivl((long long)B * H, 256)), 256, 0, stream>>>(
            W + e->w_drnnin, e->rnn_ld, e->xg_perc, W + e->w_rnnin, e->rnn_ld, B, H);
        HAB_LAUNCH_CHECK();
    }
    if (!e->rn && e->Cin == 0) {  // blind baseline policy: the recurrent encoder and the heads are all there is
        grad_tail_ready(e, e->i_wih[0]);
        return HAB_OK;
    }
    if (e->rn) { Probe pr(e, HAB_PROBE_ENC_BWD, stream); return resnet_encoder_backward(e, obs, e->last_masks, rows, B, stream); }
    // fc (Flatten -> Linear -> ReLU): d_rnnin[:, :H] already carries the ReLU mask
    { Probe pr(e, HAB_PROBE_FC_WGRAD, stream);
      HAB_TRY(linear_wgrad(dfc, e->rnn_ld, W + e->w_a3, e->fc_in, e->g(e->i_fcw), e->fc_in, B, H, e->fc_in, 32, e->fc_in / 32, 0,
                           ws, e->ws_floats, stream)); }
    HAB_TRY(colsum(dfc, e->rnn_ld, B, H, e->g(e->i_fcb), 0, ws, e->ws_floats, stream));
    grad_tail_ready(e, e->i_fcw);  // fc, recurrent encoder and heads are final; the conv stack's gradients follow
    { Probe pr(e, HAB_PROBE_FC_DGRAD, stream);
      HAB_TRY(linear_dgrad(dfc, e->rnn_ld, e->PK + e->pk_fc, e->fc_in, nullptr, 0, 0, W + e->w_da3, e->fc_in, B, e->fc_in, H, 0,
                           ws, e->ws_floats, stream)); }
    { Probe pr(e, HAB_PROBE_CONV3_DGRAD, stream);
      HAB_TRY(conv_dgrad(c3, W + e->w_da3, e->PK + e->pk_c3d, W + e->w_a2, nullptr, W + e->w_da2, ws, e->ws_floats, stream)); }
    { Probe pr(e, HAB_PROBE_CONV2_DGRAD, stream);
      HAB_TRY(conv_dgrad(c2, W + e->w_da2, e->PK + e->pk_c2d, W + e->w_a1, nullptr, W + e->w_da1, ws, e->ws_floats, stream)); }
    }
    // conv3 (no ReLU after it; its input a2 is post-ReLU -> mask on the data gradient)
    { Probe pr(e, HAB_PROBE_CONV3_WGRAD, stream);
      HAB_TRY(conv_wgrad(c3, W + e->w_a2, W + e->w_da3, e->g(e->i_c3w), e->g(e->i_c3b), ws, e->ws_floats, stream)); }
    { Probe pr(e, HAB_PROBE_CONV2_WGRAD, stream);
      HAB_TRY(conv_wgrad(c2, W + e->w_a1, W + e->w_da2, e->g(e->i_c2w), e->g(e->i_c2b), ws, e->ws_floats, stream)); }
    ObsView ov;
    ov.rgb = e->d.has_rgb ? obs->rgb : nullptr; ov.depth = e->d.has_depth ? obs->depth : nullptr; ov.rows = rows;
    ov.H = e->d.H; ov.W = e->d.W; ov.C = e->Cin;
    { Probe pr(e, HAB_PROBE_CONV1_WGRAD, stream);
      HAB_TRY(obs_conv_wgrad(c1, ov, W + e->w_da1, e->g(e->i_c1w), e->g(e->i_c1b), ws, e->ws_floats, stream)); }
    return HAB_OK;
}
// A tail exchange that failed to enqueue (event / stream-wait / ncclAllReduce error inside grad_tail_ready) is this call's error: the
// rank would otherwise go on to issue a different sequence of collectives than its peers.
extern "C" int hab_policy_backward(hab_policy* e, const hab_obs* obs, const int* rows, const int64_t* actions,
                                   const hab_pack_info* pack, const float* d_value, const float* d_log_prob,
                                   const float* d_entropy, hipStream_t stream) {
    const int rc = policy_backward_impl(e, obs, rows, actions, pack, d_value, d_log_prob, d_entropy, stream);
    if (e) e->xg_feat = e->xg_perc = nullptr;  // consumed (or refused) by this call
    return (rc == HAB_OK && e && e->comm_err != HAB_OK) ? e->comm_err : rc;
}

// Debug / test taps into the activation workspace of the last evaluate.
extern "C" int hab_policy_tap(hab_policy* e, int which, const float** ptr, int64_t* floats) {
    if (!e || !ptr || !floats || e->last_B <= 0) return HAB_ERR_ARG;
    const int64_t B = e->last_B;
    float* W = e->WK;
    if (e->rn && which != HAB_TAP_RNN_IN && which != HAB_TAP_RNN_OUT) return resnet_tap(e, which, ptr, floats);
    if (!e->rn && e->Cin == 0 && (which == HAB_TAP_CONV1 || which == HAB_TAP_CONV2 || which == HAB_TAP_CONV3)) return HAB_ERR_ARG;
    switch (which) {
        case HAB_TAP_CONV1: *ptr = W + e->w_a1; *floats = B * e->c1.Ho() * e->c1.Wo() * 32; break;
        case HAB_TAP_CONV2: *ptr = W + e->w_a2; *floats = B * e->c2.Ho() * e->c2.Wo() * 64; break;
        case HAB_TAP_CONV3: *ptr = W + e->w_a3; *floats = B * e->fc_in; break;
        case HAB_TAP_RNN_IN: *ptr = W + e->w_rnnin; *floats = B * e->rnn_ld; break;
        case HAB_TAP_RNN_OUT: *ptr = W + e->w_out[e->L - 1]; *floats = B * e->d.hidden; break;
        default: return HAB_ERR_ARG;
    }
    return HAB_OK;
}
