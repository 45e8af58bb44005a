// engine.h -- shared definition of the policy engine object (engine.hip, engine_resnet.hip).
#pragma once
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "heads.h"
#include "ops.h"
#include "problems.h"
#include "resnet_ops.h"
#include "../../include/habitat_amd.h"

struct ParamSpec {
    std::string name;
    int64_t shape[4];
    int ndim;
    int64_t offset;  // floats, 16-byte aligned
    int64_t numel;
    int is_buffer = 0;  // registered buffer of the reference module (no gradient, not an optimiser parameter)
};

struct Arena {
    int64_t used = 0;
    int64_t take(int64_t n) {
        const int64_t o = used;
        used += (n + 63) & ~(int64_t)63;  // 256-byte granules
        return o;
    }
};

struct hab_policy {
    hab_policy_desc d;
    std::vector<ParamSpec> params;
    int64_t param_floats = 0, packed_floats = 0, work_floats = 0;
    float *P = nullptr, *G = nullptr, *PK = nullptr, *WK = nullptr;
    int64_t work_bound = 0;
    int Cin = 0;
    hab::ConvDesc c1, c2, c3;  // SimpleCNN geometry (B filled per call)
    int fc_in = 0, rnn_in = 0, rnn_ld = 0, G_ = 3, L = 1;
    // param indices
    int i_c1w, i_c1b, i_c2w, i_c2b, i_c3w, i_c3b, i_fcw, i_fcb, i_aw, i_ab, i_cw, i_cb;
    int i_astd = -1;   // Gaussian head: state-independent std parameter `action_distribution.std` (or -1)
    int head_K = 0;    // Gaussian head: linear outputs (A or 2A)
    int64_t w_gsaved = -1;  // Gaussian head: [B][16] mu / std / chain factors of the last evaluate
    std::vector<int> i_wih, i_whh, i_bih, i_bhh;
    // packed offsets
    int64_t pk_c1f, pk_c2f, pk_c2d, pk_c3f, pk_c3d, pk_fc;
    int64_t pk_c1img = -1;  // conv1's bf16 weight image for the patch-resident kernel, rebuilt by every repack (-1: no image for this filter)
    std::vector<int64_t> pk_whht;
    std::vector<int64_t> pk_wiht;  // layers >= 1: W_ih transposed [H][G*H] (layer wavefront BPTT, rnn.hip); entry 0 unused (-1)
    bool last_wave = false;        // the last evaluate ran the recurrent layers as a wavefront (the backward must mirror it)
    int64_t pk_wih0 = -1;  // layer 0 W_ih, rows padded with zeros to rnn_ld floats (fused input projection of the rollout step, rnn.hip)
    // workspace offsets (floats)
    int64_t w_a1, w_a2, w_a3, w_rnnin, w_da1, w_da2, w_da3, w_drnnin, w_hinit, w_cinit, w_feat_d, w_probs, w_logitsn, w_dzv,
        w_dv, w_dfeat, w_scratch, w_ws, w_value, w_logp, w_ent, w_hmask, w_gistep, w_step_h;
    std::vector<int64_t> w_gi, w_gates, w_hn, w_hprev, w_cprev, w_c, w_out, w_dgi, w_dgh, w_dlayer;
    int64_t ws_floats = 0;
    int last_B = 0, last_n = 0;
    const uint8_t* last_masks = nullptr;
    // time-major chunked recurrence (rnn.hip, engine.hip): second stream, its events and its private scratch
    hipStream_t s2 = nullptr;
    std::vector<hipEvent_t> evs;
    int64_t w_ws2 = -1, ws2_floats = 0, w_fmask = -1, w_iota = -1;
    int last_tm = 0;  // the last evaluate ran the time-major form with this many chunks (0: packed form)
    // ResNet policy (engine_resnet.hip)
    struct ResNetPlan* rn = nullptr;
    int save_acts = 1;                                  // 0 inside act / encode: no backward follows, the fused kernels skip the saved copies
    int training = 1;                                   // nn.Module.train()/eval(): RunningMeanAndVar updates only in training
    hab_allreduce_fn allreduce_cb = nullptr;            // optional in-place all-reduce of a small device buffer (DD-PPO RMV stats)
    void* allreduce_ctx = nullptr;
    int world_size = 1;
    hab_grad_ready_fn grad_ready_cb = nullptr;          // optional: tail of the gradient arena is final (early DD-PPO all-reduce)
    void* grad_ready_ctx = nullptr;
    // device-side exchange (comm.hip): when set, the gradient tails and the RunningMeanAndVar moments are all-reduced on this RCCL
    // communicator from inside backward / forward instead of through the two callbacks above
    struct hab_comm* comm = nullptr;
    // gradients wrt `rnn_output` [B][H] / `perception_embed` [B][H] that reach the net from OUTSIDE the fused heads: auxiliary losses
    // computed by torch modules on the autograd bridge (rl/ppo/policy.py:386-394); consumed and cleared by the next hab_policy_backward
    const float* xg_feat = nullptr; const float* xg_perc = nullptr;
    int comm_err = 0;                                   // first error of a tail exchange of the current backward (HAB_OK: none)
    int64_t comm_first = -1;                            // grads[comm_first ..) has been enqueued for exchange in this backward (-1: nothing)
    hipStream_t cur_stream = nullptr;                   // stream of the running hab_policy_backward
    // probe
    uint64_t probe_mask = 0;  // bit t set: call site HAB_PROBE_<t> is bracketed by HIP events on the launch stream
    struct ProbeEv { int tag; hipEvent_t first, second; };
    std::vector<ProbeEv> probe_events;
    size_t probe_used = 0;
    double probe_flops[64] = {0}, probe_bytes[64] = {0};  // algorithmic work of the bracketed call sites that report it

    float* p(int i) const { return P + params[i].offset; }
    float* g(int i) const { return G + params[i].offset; }
};

inline int add_param(hab_policy* e, const std::string& name, std::initializer_list<int64_t> shape) {
    ParamSpec s;
    s.name = name;
    s.ndim = (int)shape.size();
    s.numel = 1;
    int k = 0;
    for (auto v : shape) { s.shape[k++] = v; s.numel *= v; }
    for (; k < 4; ++k) s.shape[k] = 1;
    s.offset = e->param_floats;
    e->param_floats += (s.numel + 3) & ~(int64_t)3;
    e->params.push_back(s);
    return (int)e->params.size() - 1;
}


int tm_chunks_cfg();         // time chunks of the time-major recurrence (HAB_RNN_CHUNKS; 0 / 1: packed form / one chunk)
int tm_chunks_resnet_cfg(int frames);  // ... for the ResNet policies (HAB_RNN_CHUNKS_RESNET; default: 2 chunks from 4096 frames on, else packed)
int build_resnet(hab_policy* e);
void destroy_resnet(hab_policy* e);
int resnet_repack(hab_policy* e, hipStream_t s);
void grad_tail_ready(hab_policy* e, int first_param);
int comm_exchange_async(struct hab_comm* c, float* buf, int64_t first, int64_t count, hipStream_t compute);
extern "C" int hab_comm_allreduce_sum(struct hab_comm* c, float* buf, int64_t count, hipStream_t stream);
int resnet_feature_shape(const hab_policy* e, int* c, int* hf, int* wf);
int resnet_encode(hab_policy* e, const hab_obs* obs, int n, float* out, hipStream_t s);
int resnet_encoder_forward(hab_policy* e, const hab_obs* obs, const uint8_t* masks, const int* rows, int B, hipStream_t s, int f0 = 0, int nB = -1);
int resnet_encoder_backward(hab_policy* e, const hab_obs* obs, const uint8_t* masks, const int* rows, int B, hipStream_t s);
int resnet_tap(hab_policy* e, int which, const float** ptr, int64_t* floats);

struct Probe {
    hab_policy* e; hipStream_t s; bool on;
    Probe(hab_policy* e_, int tag, hipStream_t s_, double flops = 0.0, double bytes = 0.0) : e(e_), s(s_), on((e_->probe_mask >> tag) & 1) {
        if (!on) return;
        e->probe_flops[tag] += flops; e->probe_bytes[tag] += bytes;
        if (e->probe_used == e->probe_events.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            e->probe_events.push_back({tag, a, b});
        }
        e->probe_events[e->probe_used].tag = tag;
        (void)hipEventRecord(e->probe_events[e->probe_used].first, s);
    }
    ~Probe() {
        if (!on) return;
        (void)hipEventRecord(e->probe_events[e->probe_used].second, s);
        e->probe_used++;
    }
};

