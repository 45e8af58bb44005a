// engine_resnet.hip -- the GroupNorm-ResNet visual encoder of PointNavResNetPolicy inside the policy engine:
// layer program (stem, BasicBlock / Bottleneck stages, compression, visual_fc, goal + previous-action embeddings),
// its parameter table in the reference's state_dict() order, forward with saved activations and the full backward.
//
// Replaces (habitat-baselines/habitat_baselines/rl/ddppo/policy/): resnet_policy.py:162-276 (ResNetEncoder),
// resnet.py:37-69,116-152,196-281 (blocks, ResNet), running_mean_and_var.py:24-78, resnet_policy.py:625-767
// (PointNavResNetNet.forward: visual_fc, tgt_embeding, prev_action_embedding, concat) and the autograd backward
// through them (rl/ppo/ppo.py:253).
//
// Dataflow per frame batch (all NHWC fp32, contractions on the igemm kernel, the rest in resnet_ops.hip):
//   obs --ingest(uint8*1/255, avg_pool2d(2), concat)--> x0 --RunningMeanAndVar--> x0
//   x0 --conv7x7/2--> raw --GN+ReLU--> n0 --maxpool3x3/2--> pool --blocks--> ... --conv3x3--> raw --GN(1)+ReLU--> comp
//   comp --Flatten(NCHW order)+Linear+ReLU--> rnn_in[:, :H];  embeddings --> rnn_in[:, H:H+64]
// Block: conv_i -> raw_i -> GN(+ReLU) -> n_i ... last: GN + residual + ReLU -> out.  raw_i (GroupNorm input) and
// n_i (next conv's input) are both kept for the backward pass, exactly what autograd keeps in the reference.
#include "engine.h"

using namespace hab;

namespace {

struct RnConv {
    ConvDesc cd;                 // B = 0 until a call
    int groups = 0;
    int i_w = -1, i_gamma = -1, i_beta = -1;
    int64_t pk_f = -1, pk_d = -1;   // packed weights (pk_d < 0: no data gradient needed)
    int64_t pk_p = -1;              // forward weights as three bf16 planes for the fused small-batch kernel (conv_gn_slab.h), or -1
    int64_t w_raw = -1, w_mean = -1, w_rstd = -1, w_out = -1;
    int cgroups = 1;             // ResNeXt: groups of the 3x3 convolution (weight parameter is (Cout, C / cgroups, 3, 3))
    int64_t out_floats() const { return (int64_t)cd.Ho() * cd.Wo() * cd.Cout; }
    int64_t in_floats() const { return (int64_t)cd.H * cd.W * cd.C; }
};

struct RnBlock {
    std::vector<int> convs;  // main branch
    int ds = -1;             // downsample conv or -1
    int64_t w_in = -1;       // block input (previous block's out / pool)
    int64_t w_out = -1;      // block output
    // squeeze-and-excitation gate (SEBottleneck, resnet.py:155-187): out = relu(se(convs(x)) * convs(x) + identity)
    int se_c = 0, se_r = 0;  // channels, reduced width (channels / 16); 0 = no SE
    int i_se1w = -1, i_se1b = -1, i_se2w = -1, i_se2b = -1;
    int64_t w_se_pool = -1, w_se_h = -1, w_se_gate = -1;   // [B][C], [B][C/16], [B][C] (the last conv's w_out holds the un-gated GN output)
};

}  // namespace

struct ResNetPlan {
    // blind: no visual sensor reaches the net (`force_blind_policy`, or an observation space without images: resnet_policy.py:553-592,
    // ResNetEncoder.is_blind :249-251) -- no backbone, no compression, no visual_fc; the recurrent encoder's input is the embeddings alone
    bool blind = false;
    int cpad = 4, creal = 4, H2 = 0, W2 = 0;
    int64_t w_x0 = -1;
    RnConv stem;
    int64_t w_pool = -1, w_pool_idx = -1;
    int poolH = 0, poolW = 0;
    std::vector<RnConv> convs;
    std::vector<RnBlock> blocks;
    RnConv comp;
    int i_embb = -1;
    int i_fcw = -1, i_fcb = -1, i_emb = -1, i_tgw = -1, i_tgb = -1, i_mean = -1, i_var = -1, i_count = -1;
    int i_objw = -1, i_gpsw = -1, i_gpsb = -1, i_cmpw = -1, i_cmpb = -1, i_pgw = -1, i_pgb = -1, i_pxw = -1, i_pxb = -1;
    int c_rgb = -1, c_depth = -1, c_sem = -1;  // first channel of each visual key in the concatenated encoder input
    int nslots = 0;
    int64_t pk_fc = -1;
    int fc_in = 0, comp_c = 0, comp_hw = 0, comp_fh = 0, comp_fw = 0;
    // gradient scratch
    int64_t w_gstem[2] = {-1, -1};
    std::vector<int64_t> w_gbuf;
    int64_t gbuf_floats = 0;  // per frame
    int64_t w_chansums = -1;  // [B][2][Cmax]
    int64_t w_stats = -1;     // RunningMeanAndVar batch moments: mean[8], count, var[8] at +16 (+ padding)
    int64_t w_dscratch = -1;  // doubles for chan_moment partials
    int64_t w_embsave = -1;   // [B][4] goal features + previous-action token of the last forward
    int64_t w_se_scr = -1;    // SE backward scratch: dgate / dz / dpool [B][Cmax] x 3 + dh [B][Cmax/16]
    int64_t w_gdense = -1;    // dense OIHW weight gradient of the largest grouped convolution
    int64_t gdense_floats = 0;
    int cmax = 0;
    int nlayers[4] = {2, 2, 2, 2};
    // stem of the last forward: GroupNorm + ReLU + max-pool ran fused (the normalised stem output was not stored); stem_act_valid: it
    // has been materialised since (a debug tap asked for it) -- the backward then takes the unfused form, which reads it as the ReLU
    // mask (so that activations patched through the tap are honoured, oracle/parity.py::MaskInjector)
    bool stem_fused = false, stem_act_valid = true;
    // training-mode RunningMeanAndVar with both stem kernels of the strip family: the ingest leaves x0 UN-normalised (it accumulates the batch
    // moments in the same pass) and the stem forward / weight gradient apply the per-channel affine while staging (w_stats + 32: 16 floats).
    // x0_raw: x0 of the last forward is in that state (a debug tap of the encoder input normalises it in place and clears the flag).
    bool stem_takes_raw = false, x0_raw = false;
};

static int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

static int add_conv_gn(hab_policy* e, RnConv& c, const std::string& wname, const std::string& gnname, int cin_real) {
    c.i_w = add_param(e, wname + ".weight", {c.cd.Cout, cin_real, c.cd.KH, c.cd.KW});
    c.i_gamma = add_param(e, gnname + ".weight", {c.cd.Cout});
    c.i_beta = add_param(e, gnname + ".bias", {c.cd.Cout});
    return HAB_OK;
}

int build_resnet(hab_policy* e) {
    const hab_policy_desc& d = e->d;
    if (d.backbone != HAB_BACKBONE_RESNET18 && d.backbone != HAB_BACKBONE_RESNET50 && d.backbone != HAB_BACKBONE_RESNEXT50 &&
        d.backbone != HAB_BACKBONE_SE_RESNET50 && d.backbone != HAB_BACKBONE_SE_RESNEXT50 && d.backbone != HAB_BACKBONE_SE_RESNEXT101)
        return HAB_ERR_UNSUPPORTED;
    if (d.baseplanes <= 0 || d.baseplanes % 8) return HAB_ERR_UNSUPPORTED;
    const bool blind = !d.has_rgb && !d.has_depth && !d.has_semantic;
    if (!blind && (d.H < 2 || d.W < 2)) return HAB_ERR_UNSUPPORTED;  // odd sizes: avg_pool2d(2) floors, the last row / column is dropped (F.avg_pool2d)
    if (d.rnn_type != HAB_RNN_GRU && d.rnn_type != HAB_RNN_LSTM) return HAB_ERR_ARG;
    if (d.goal_dim != 0 && d.goal_dim != 2) return HAB_ERR_UNSUPPORTED;  // 2-D polar pointgoal (resnet_policy.py:662-672)
    ResNetPlan* r = new ResNetPlan();
    e->rn = r;
    const int H = d.hidden;
    e->G_ = d.rnn_type == HAB_RNN_GRU ? 3 : 4;
    e->L = d.rnn_layers;
    r->creal = (d.has_rgb ? 3 : 0) + (d.has_depth ? 1 : 0) + (d.has_semantic ? 1 : 0);
    r->blind = blind;
    e->Cin = r->creal;
    r->cpad = r->creal <= 4 ? 4 : 8;
    if (!blind) {   // channel offsets from the observation-space key order
        int off = 0, seen = 0;
        int order = d.visual_order ? d.visual_order : (1 | (2 << 2) | (3 << 4));
        for (int q = 0; q < 3; ++q) {
            const int key = (order >> (2 * q)) & 3;
            if (key == 1 && d.has_rgb && r->c_rgb < 0) { r->c_rgb = off; off += 3; ++seen; }
            else if (key == 2 && d.has_depth && r->c_depth < 0) { r->c_depth = off; off += 1; ++seen; }
            else if (key == 3 && d.has_semantic && r->c_sem < 0) { r->c_sem = off; off += 1; ++seen; }
        }
        if (off != r->creal || seen != (d.has_rgb ? 1 : 0) + (d.has_depth ? 1 : 0) + (d.has_semantic ? 1 : 0)) return HAB_ERR_ARG;
    }
    r->H2 = d.H / 2; r->W2 = d.W / 2;
    const int bp = d.baseplanes, ng = bp / 2;
    // resnet.py:296-345: block type, stage depths, ResNeXt (expansion 2, base width x2, grouped 3x3 with cardinality
    // base_planes / 2 -- in the FIRST block of every stage only, _make_layer :257-268 passes it to no other), SE gate
    const bool bottleneck = d.backbone != HAB_BACKBONE_RESNET18;
    const bool resnext = d.backbone == HAB_BACKBONE_RESNEXT50 || d.backbone == HAB_BACKBONE_SE_RESNEXT50 || d.backbone == HAB_BACKBONE_SE_RESNEXT101;
    const bool se = d.backbone == HAB_BACKBONE_SE_RESNET50 || d.backbone == HAB_BACKBONE_SE_RESNEXT50 || d.backbone == HAB_BACKBONE_SE_RESNEXT101;
    const int expansion = !bottleneck ? 1 : (resnext ? 2 : 4);
    const int cardinality = resnext ? bp / 2 : 1;
    static const int L18[4] = {2, 2, 2, 2}, L50[4] = {3, 4, 6, 3}, L101[4] = {3, 4, 23, 3};
    const int* layers = !bottleneck ? L18 : (d.backbone == HAB_BACKBONE_SE_RESNEXT101 ? L101 : L50);
    r->nlayers[0] = layers[0]; r->nlayers[1] = layers[1]; r->nlayers[2] = layers[2]; r->nlayers[3] = layers[3];

    // ---- parameter table in the reference's state_dict order (resnet_policy.py:389-396,454-456 first) ----
    const bool gauss = d.action_dist == HAB_DIST_GAUSSIAN;
    if (gauss && (d.num_actions > 4)) return HAB_ERR_UNSUPPORTED;
    if (gauss) {  // nn.Linear(num_actions, 32) on masks * prev_actions (resnet_policy.py:424-428)
        r->i_emb = add_param(e, "net.prev_action_embedding.weight", {32, d.num_actions});
        r->i_embb = add_param(e, "net.prev_action_embedding.bias", {32});
    } else {
        r->i_emb = add_param(e, "net.prev_action_embedding.weight", {d.num_actions + 1, 32});
    }
    r->nslots = 1;
    if (d.goal_dim == 2) {
        r->i_tgw = add_param(e, "net.tgt_embeding.weight", {32, 3});
        r->i_tgb = add_param(e, "net.tgt_embeding.bias", {32});
        ++r->nslots;
    }
    if (d.num_object_categories > 0) { r->i_objw = add_param(e, "net.obj_categories_embedding.weight", {d.num_object_categories, 32}); ++r->nslots; }
    if (d.has_gps) {
        r->i_gpsw = add_param(e, "net.gps_embedding.weight", {32, 2});
        r->i_gpsb = add_param(e, "net.gps_embedding.bias", {32});
        ++r->nslots;
    }
    // (resnet_policy.py:489-515: pointgoal_embedding, [heading_embedding], proximity_embedding sit between gps and compass)
    if (d.pointgoal_dim < 0 || d.pointgoal_dim > 4 || d.proximity_dim < 0 || d.proximity_dim > 4) return HAB_ERR_UNSUPPORTED;
    if (d.pointgoal_dim > 0) {
        r->i_pgw = add_param(e, "net.pointgoal_embedding.weight", {32, d.pointgoal_dim});
        r->i_pgb = add_param(e, "net.pointgoal_embedding.bias", {32});
        ++r->nslots;
    }
    if (d.proximity_dim > 0) {
        r->i_pxw = add_param(e, "net.proximity_embedding.weight", {32, d.proximity_dim});
        r->i_pxb = add_param(e, "net.proximity_embedding.bias", {32});
        ++r->nslots;
    }
    if (d.has_compass) {
        r->i_cmpw = add_param(e, "net.compass_embedding.weight", {32, 2});
        r->i_cmpb = add_param(e, "net.compass_embedding.bias", {32});
        ++r->nslots;
    }
    const std::string ve = "net.visual_encoder.";
    if (d.normalize_visual_inputs && !blind) {
        r->i_mean = add_param(e, ve + "running_mean_and_var._mean", {1, r->creal, 1, 1});
        r->i_var = add_param(e, ve + "running_mean_and_var._var", {1, r->creal, 1, 1});
        r->i_count = add_param(e, ve + "running_mean_and_var._count", {});
        e->params[r->i_mean].is_buffer = e->params[r->i_var].is_buffer = e->params[r->i_count].is_buffer = 1;
    }
    const std::string bb = ve + "backbone.";
    int inplanes = bp, curH = 0, curW = 0;
    if (!blind) {
    r->stem.cd = ConvDesc{0, r->H2, r->W2, r->cpad, bp, 7, 7, 2, 3};
    r->stem.cd.Creal = r->creal;
    r->stem.groups = ng;
    add_conv_gn(e, r->stem, bb + "conv1.0", bb + "conv1.1", r->creal);
    const int sh = r->stem.cd.Ho(), sw = r->stem.cd.Wo();
    r->poolH = conv_out(sh, 3, 2, 1); r->poolW = conv_out(sw, 3, 2, 1);
    curH = r->poolH; curW = r->poolW;
    for (int li = 0; li < 4; ++li) {
        const int planes = (resnext ? 2 * bp : bp) << li;
        for (int bi = 0; bi < layers[li]; ++bi) {
            const int stride = (bi == 0 && li > 0) ? 2 : 1;
            const std::string bpfx = bb + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            const bool has_ds = bi == 0 && (stride != 1 || inplanes != planes * expansion);
            RnBlock blk;
            auto push = [&](int cin, int cout, int k, int s, int p, int hh, int ww, const std::string& w, const std::string& g,
                            int cgroups = 1) {
                RnConv c;
                c.cd = ConvDesc{0, hh, ww, cin, cout, k, k, s, p};
                c.groups = ng;
                c.cgroups = cgroups;
                add_conv_gn(e, c, w, g, cin / cgroups);
                if (cgroups > 1) r->gdense_floats = std::max<int64_t>(r->gdense_floats, (int64_t)cout * cin * k * k);
                r->convs.push_back(c);
                return (int)r->convs.size() - 1;
            };
            const int oh = conv_out(curH, 3, stride, 1), ow = conv_out(curW, 3, stride, 1);
            if (!bottleneck) {
                blk.convs.push_back(push(inplanes, planes, 3, stride, 1, curH, curW, bpfx + "convs.0", bpfx + "convs.1"));
                blk.convs.push_back(push(planes, planes, 3, 1, 1, oh, ow, bpfx + "convs.3", bpfx + "convs.4"));
            } else {
                blk.convs.push_back(push(inplanes, planes, 1, 1, 0, curH, curW, bpfx + "convs.0", bpfx + "convs.1"));
                blk.convs.push_back(push(planes, planes, 3, stride, 1, curH, curW, bpfx + "convs.3", bpfx + "convs.4",
                                         bi == 0 ? cardinality : 1));
                blk.convs.push_back(push(planes, planes * expansion, 1, 1, 0, oh, ow, bpfx + "convs.6", bpfx + "convs.7"));
            }
            if (has_ds) blk.ds = push(inplanes, planes * expansion, 1, stride, 0, curH, curW, bpfx + "downsample.0", bpfx + "downsample.1");
            if (se) {  // registered after convs / downsample (SEBottleneck.__init__, resnet.py:166-176)
                blk.se_c = planes * expansion; blk.se_r = blk.se_c / 16;
                if (blk.se_r < 4 || (blk.se_r & 3)) return HAB_ERR_UNSUPPORTED;
                blk.i_se1w = add_param(e, bpfx + "se.excite.0.weight", {blk.se_r, blk.se_c});
                blk.i_se1b = add_param(e, bpfx + "se.excite.0.bias", {blk.se_r});
                blk.i_se2w = add_param(e, bpfx + "se.excite.2.weight", {blk.se_c, blk.se_r});
                blk.i_se2b = add_param(e, bpfx + "se.excite.2.bias", {blk.se_c});
            }
            r->blocks.push_back(blk);
            inplanes = planes * expansion;
            curH = oh; curW = ow;
        }
    }
    // compression (resnet_policy.py:213-234)
    const int fh = (int)ceil(r->H2 / 32.0), fw = (int)ceil(r->W2 / 32.0);
    if (fh != curH || fw != curW) return HAB_ERR_UNSUPPORTED;
    const int ncomp = (int)lrint(2048.0 / (fh * fw));
    if (ncomp <= 0 || ncomp % 4) return HAB_ERR_UNSUPPORTED;
    r->comp.cd = ConvDesc{0, curH, curW, inplanes, ncomp, 3, 3, 1, 1};
    r->comp.groups = 1;
    add_conv_gn(e, r->comp, ve + "compression.0", ve + "compression.1", inplanes);
    r->comp_c = ncomp; r->comp_hw = fh * fw; r->comp_fh = fh; r->comp_fw = fw; r->fc_in = ncomp * fh * fw;
    r->i_fcw = add_param(e, "net.visual_fc.1.weight", {H, r->fc_in});
    r->i_fcb = add_param(e, "net.visual_fc.1.bias", {H});
    }  // !blind
    e->fc_in = r->fc_in;
    e->rnn_in = (blind ? 0 : H) + 32 * r->nslots;
    e->rnn_ld = (e->rnn_in + 15) & ~15;
    if (e->rnn_ld != e->rnn_in) return HAB_ERR_UNSUPPORTED;  // (32-wide embedding slots behind a hidden size % 64 == 0: never padded)
    const std::string rn = "net.state_encoder.rnn.";
    for (int l = 0; l < d.rnn_layers; ++l) {
        const int in = l == 0 ? e->rnn_in : H;
        const std::string sfx = "_l" + std::to_string(l);
        e->i_wih.push_back(add_param(e, rn + "weight_ih" + sfx, {e->G_ * H, in}));
        e->i_whh.push_back(add_param(e, rn + "weight_hh" + sfx, {e->G_ * H, H}));
        e->i_bih.push_back(add_param(e, rn + "bias_ih" + sfx, {e->G_ * H}));
        e->i_bhh.push_back(add_param(e, rn + "bias_hh" + sfx, {e->G_ * H}));
    }
    if (gauss) {  // GaussianNet (utils/common.py:124-149): the std parameter is registered before the linear layer
        if (d.gauss_flags & HAB_GAUSS_USE_STD_PARAM) e->i_astd = add_param(e, "action_distribution.std", {d.num_actions});
        e->head_K = (d.gauss_flags & HAB_GAUSS_USE_STD_PARAM) ? d.num_actions : 2 * d.num_actions;
        e->i_aw = add_param(e, "action_distribution.mu_maybe_std.weight", {e->head_K, H});
        e->i_ab = add_param(e, "action_distribution.mu_maybe_std.bias", {e->head_K});
    } else {
        e->i_aw = add_param(e, "action_distribution.linear.weight", {d.num_actions, H});
        e->i_ab = add_param(e, "action_distribution.linear.bias", {d.num_actions});
    }
    e->i_cw = add_param(e, "critic.fc.weight", {1, H});
    e->i_cb = add_param(e, "critic.fc.bias", {1});

    // ---- packed weights ----
    Arena pk;
    auto pack_conv = [&](RnConv& c, bool need_d) {
        const int64_t n = (int64_t)c.cd.Cout * c.cd.KH * c.cd.KW * c.cd.C;
        c.pk_f = pk.take(n);
        if (need_d) c.pk_d = pk.take(n);
        if (conv_gn_fused_ok(c.cd.C, c.cd.Cout, c.cd.H, c.cd.W, c.cd.KH, c.cd.KW, c.cd.stride, c.cd.pad, c.groups))
            c.pk_p = pk.take((3 * n + 1) / 2);  // 3 planes x 2 bytes per weight
    };
    if (!blind) {
    pack_conv(r->stem, false);
    if (stem_conv_ok(r->stem.cd.H, r->stem.cd.W, r->stem.cd.C, r->stem.cd.Cout, 7, 7, 2, 3)) r->stem.pk_p = pk.take(STEM_PLANE_FLOATS);
    r->stem_takes_raw = r->stem.pk_p >= 0 && r->cpad == 4 && r->stem.cd.Cout == 32 &&
                        stem_wgrad_ok(r->stem.cd.H, r->stem.cd.W, r->stem.cd.C, r->stem.cd.Cout, 7, 7, 2, 3);
    for (auto& c : r->convs) pack_conv(c, true);
    pack_conv(r->comp, true);
    r->pk_fc = pk.take((int64_t)H * r->fc_in);
    }
    for (int l = 0; l < d.rnn_layers; ++l) e->pk_whht.push_back(pk.take((int64_t)e->G_ * H * H));
    for (int l = 0; l < d.rnn_layers; ++l) e->pk_wiht.push_back(l == 0 ? -1 : pk.take((int64_t)e->G_ * H * H));
    e->pk_wih0 = pk.take((int64_t)e->G_ * H * e->rnn_ld);
    e->packed_floats = pk.used;

    // ---- workspace ----
    Arena wk;
    const int64_t B = d.max_frames, F = d.max_frames;
    if (!blind) r->w_x0 = wk.take(B * r->H2 * r->W2 * r->cpad);
    auto place = [&](RnConv& c, bool own_out) {
        c.w_raw = wk.take(B * c.out_floats());
        c.w_mean = wk.take(B * c.groups);
        c.w_rstd = wk.take(B * c.groups);
        if (own_out) c.w_out = wk.take(B * c.out_floats());
        r->cmax = std::max(r->cmax, c.cd.Cout);
    };
    if (!blind) {
    place(r->stem, true);
    r->w_pool = wk.take(B * r->poolH * r->poolW * bp);
    r->w_pool_idx = wk.take((B * r->poolH * r->poolW * bp + 3) / 4);
    int64_t prev = r->w_pool;
    r->gbuf_floats = (int64_t)r->poolH * r->poolW * bp;
    for (auto& blk : r->blocks) {
        blk.w_in = prev;
        for (size_t q = 0; q < blk.convs.size(); ++q) {
            RnConv& c = r->convs[blk.convs[q]];
            place(c, true);  // last conv's w_out is the block output
            r->gbuf_floats = std::max(r->gbuf_floats, c.out_floats());
        }
        if (blk.ds >= 0) { place(r->convs[blk.ds], true); r->gbuf_floats = std::max(r->gbuf_floats, r->convs[blk.ds].out_floats()); }
        blk.w_out = r->convs[blk.convs.back()].w_out;
        if (blk.se_c) {  // the gated block output needs its own buffer: the last conv's w_out keeps the un-gated GN output
            blk.w_out = wk.take(B * r->convs[blk.convs.back()].out_floats());
            blk.w_se_pool = wk.take(B * blk.se_c); blk.w_se_h = wk.take(B * blk.se_r); blk.w_se_gate = wk.take(B * blk.se_c);
        }
        prev = blk.w_out;
    }
    place(r->comp, true);
    r->gbuf_floats = std::max(r->gbuf_floats, r->comp.out_floats());
    for (int k = 0; k < 2; ++k) r->w_gstem[k] = wk.take(B * r->stem.out_floats());
    for (int k = 0; k < 6; ++k) r->w_gbuf.push_back(wk.take(B * r->gbuf_floats));
    r->w_chansums = wk.take(B * 2 * r->cmax);
    r->w_stats = wk.take(64);
    r->w_dscratch = wk.take(2 * (INGEST_MOM_MAX_BLOCKS * 16 + 16));  // doubles: per-workgroup moment partials of the ingest + their totals
    if (se) r->w_se_scr = wk.take(B * ((int64_t)3 * r->cmax + r->cmax / 16 + 16));
    if (r->gdense_floats) r->w_gdense = wk.take(r->gdense_floats);
    }  // !blind
    r->w_embsave = wk.take(B * 4 * EMB_MAX_SLOTS);
    // shared tail (RNN, heads) -- same layout as the SimpleCNN engine
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
    if (gauss) e->w_gsaved = wk.take(B * 16);
    e->w_hmask = wk.take((int64_t)2 * d.rnn_layers * d.max_envs * H);
    e->w_gistep = wk.take((int64_t)d.max_envs * e->G_ * H);
    e->w_step_h = wk.take((int64_t)d.max_envs * H * 2);
    e->ws_floats = (int64_t)32 << 20;
    e->w_ws = wk.take(e->ws_floats);
    // second stream of the time-major chunked recurrence (engine.hip): its own split-K scratch (same cap: the split-K plans, hence the
    // bits, must not depend on the stream), the dense per-frame episode-start mask, an iota
    // -- only when that form is selected (default for ResNet policies: packed; measured slower, engine.hip): every ResNet engine, incl. each
    // VER inference worker's private one, would otherwise carry 128 MB + 5 bytes per frame for nothing (w_ws2 < 0 keeps the packed form)
    if (tm_chunks_resnet_cfg((int)B) > 0) {  // (B = max_frames: the form this engine can ever select)
        e->ws2_floats = e->ws_floats;
        e->w_ws2 = wk.take(e->ws2_floats);
        e->w_fmask = wk.take((B + 3) / 4 + 64);
        e->w_iota = wk.take(B + 64);
    }
    e->work_floats = wk.used;
    return HAB_OK;
}

void destroy_resnet(hab_policy* e) {
    delete e->rn;
    e->rn = nullptr;
}

int resnet_repack(hab_policy* e, hipStream_t s) {
    ResNetPlan* r = e->rn;
    const int H = e->d.hidden;
    auto rp = [&](const RnConv& c, int cin_real) {
        if (c.cgroups > 1)
            return repack_conv_grouped(e->p(c.i_w), e->PK + c.pk_f, c.pk_d >= 0 ? e->PK + c.pk_d : nullptr, c.cd.Cout, c.cd.C, c.cgroups,
                                       c.cd.KH, c.cd.KW, s);
        return repack_conv(e->p(c.i_w), e->PK + c.pk_f, c.pk_d >= 0 ? e->PK + c.pk_d : nullptr, c.cd.Cout, cin_real, c.cd.KH, c.cd.KW,
                           c.cd.C, s);
    };
    auto planes = [&](const RnConv& c) {
        if (c.pk_p < 0) return (int)HAB_OK;
        return weight_planes(e->PK + c.pk_f, c.cd.Cout, c.cd.KH * c.cd.KW * c.cd.C, reinterpret_cast<unsigned short*>(e->PK + c.pk_p), s);
    };
    if (!r->blind) {
        HAB_TRY(rp(r->stem, r->creal));
        if (r->stem.pk_p >= 0) HAB_TRY(stem_weight_planes(e->PK + r->stem.pk_f, reinterpret_cast<unsigned short*>(e->PK + r->stem.pk_p), s));
        for (const auto& c : r->convs) { HAB_TRY(rp(c, c.cd.C)); HAB_TRY(planes(c)); }
        HAB_TRY(rp(r->comp, r->comp.cd.C));
        HAB_TRY(planes(r->comp));
        HAB_TRY(repack_flatten(e->p(r->i_fcw), e->PK + r->pk_fc, H, r->comp_c, r->comp_hw, s));
    }
    for (int l = 0; l < e->L; ++l) HAB_TRY(transpose2d(e->p(e->i_whh[l]), e->PK + e->pk_whht[l], e->G_ * H, H, s));
    for (int l = 1; l < e->L; ++l) HAB_TRY(transpose2d(e->p(e->i_wih[l]), e->PK + e->pk_wiht[l], e->G_ * H, H, s));
    HAB_TRY(pad_rows(e->p(e->i_wih[0]), e->PK + e->pk_wih0, e->G_ * H, e->rnn_in, e->rnn_ld, s));
    return HAB_OK;
}

// Embedding slots in the order PointNavResNetNet.forward concatenates them (resnet_policy.py:662-755):
// pointgoal_with_gps_compass, pointgoal, proximity, objectgoal, compass, gps, previous action.
static int fill_embed_slots(hab_policy* e, const hab_obs* obs, EmbedSlot* sl, bool grads) {
    ResNetPlan* r = e->rn;
    const hab_policy_desc& d = e->d;
    int n = 0;
    auto add = [&](int kind, const void* in, int iw, int ib, int ntok) {
        EmbedSlot& q = sl[n++];
        q.kind = kind; q.in = in; q.w = e->p(iw); q.b = ib >= 0 ? e->p(ib) : nullptr; q.ntok = ntok;
        q.dw = grads ? e->g(iw) : nullptr; q.db = (grads && ib >= 0) ? e->g(ib) : nullptr;
        return in != nullptr;
    };
    bool ok = true;
    if (d.goal_dim == 2) ok &= add(EMB_POLAR, obs->goal, r->i_tgw, r->i_tgb, 0);
    if (d.pointgoal_dim > 0) ok &= add(EMB_LINN, obs->pointgoal, r->i_pgw, r->i_pgb, d.pointgoal_dim);
    if (d.proximity_dim > 0) ok &= add(EMB_LINN, obs->proximity, r->i_pxw, r->i_pxb, d.proximity_dim);
    if (d.num_object_categories > 0) ok &= add(EMB_TOKEN, obs->objectgoal, r->i_objw, -1, d.num_object_categories);
    if (d.has_compass) ok &= add(EMB_COSSIN, obs->compass, r->i_cmpw, r->i_cmpb, 0);
    if (d.has_gps) ok &= add(EMB_LIN2, obs->gps, r->i_gpsw, r->i_gpsb, 0);
    if (d.action_dist == HAB_DIST_GAUSSIAN) ok &= add(EMB_PREVLIN, obs->prev_actions, r->i_emb, r->i_embb, d.num_actions);
    else ok &= add(EMB_PREV, obs->prev_actions, r->i_emb, -1, d.num_actions + 1);
    return (ok && n == r->nslots) ? HAB_OK : HAB_ERR_ARG;
}

// conv -> raw, GroupNorm (+residual, +ReLU) -> out.  f0: first frame of the workspace buffers this call works on (time-major chunks of
// a minibatch, engine.hip); `in` / `residual` are already offset by the caller.
static int conv_gn_forward(hab_policy* e, const RnConv& c, const float* in, const float* residual, int relu, int B, hipStream_t s, int f0 = 0) {
    float* W = e->WK;
    ConvDesc cd = c.cd;
    cd.B = B;
    float* raw = W + c.w_raw + (int64_t)f0 * c.out_floats();
    float* out = W + c.w_out + (int64_t)f0 * c.out_floats();
    float* mean = W + c.w_mean + (int64_t)f0 * c.groups;
    float* rstd = W + c.w_rstd + (int64_t)f0 * c.groups;
    if (&c == &e->rn->stem && c.pk_p >= 0) {  // stem: input strip resident in LDS (stem_conv_strip.h)
        const int rc = stem_conv_forward(in, reinterpret_cast<const unsigned short*>(e->PK + c.pk_p), raw, B, cd.H, cd.W, s,
                                         e->rn->x0_raw ? W + e->rn->w_stats + 32 : nullptr);
        if (rc == 1 && e->rn->x0_raw) return HAB_ERR_UNSUPPORTED;
        if (rc != 0 && rc != 1) return rc;
        if (rc == 1) HAB_TRY(conv_fwd(cd, in, e->PK + c.pk_f, nullptr, raw, 0, W + e->w_ws, e->ws_floats, s));
        GnArgs g;
        g.x = raw; g.y = out; g.gamma = e->p(c.i_gamma); g.beta = e->p(c.i_beta); g.residual = residual;
        g.mean = mean; g.rstd = rstd; g.B = B; g.HW = cd.Ho() * cd.Wo(); g.C = cd.Cout; g.groups = c.groups;
        g.relu = relu; g.eps = 1e-5f; g.scratch = W + e->w_ws; g.scratch_floats = e->ws_floats;
        return groupnorm_forward(g, s);
    }
    // the rollout's act / encode (small batches, nothing kept for a backward): convolution + GroupNorm in one launch (conv_gn_slab.h).
    // Small evaluate minibatches keep the unfused pair -- the kernels every update-sized minibatch runs, so that the golden update tests
    // exercise the production learner path (the fused kernels can also write the pre-normalisation output and the statistics a backward
    // pass needs: tests/test_gpu_conv_gn.py).
    constexpr int cgs_max_b = 256;
    if (c.pk_p >= 0 && B <= cgs_max_b && !e->save_acts) {
        ConvGnArgs q;
        q.x = in; q.w_planes = reinterpret_cast<const unsigned short*>(e->PK + c.pk_p); q.gamma = e->p(c.i_gamma); q.beta = e->p(c.i_beta);
        q.residual = residual; q.y = out;
        q.B = B; q.H = cd.H; q.W = cd.W; q.C = cd.C; q.Cout = cd.Cout; q.KH = cd.KH; q.KW = cd.KW; q.stride = cd.stride; q.pad = cd.pad;
        q.groups = c.groups; q.relu = relu; q.eps = 1e-5f;
        const int rc = conv_gn_fused(q, s);
        if (rc != 1) return rc;
    }
    HAB_TRY(conv_fwd(cd, in, e->PK + c.pk_f, nullptr, raw, 0, W + e->w_ws, e->ws_floats, s));
    GnArgs g;
    g.x = raw; g.y = out; g.gamma = e->p(c.i_gamma); g.beta = e->p(c.i_beta); g.residual = residual;
    g.mean = mean; g.rstd = rstd; g.B = B; g.HW = cd.Ho() * cd.Wo(); g.C = cd.Cout; g.groups = c.groups;
    g.relu = relu; g.eps = 1e-5f;
    g.scratch = W + e->w_ws; g.scratch_floats = e->ws_floats;  // chunk-parallel statistics for frames > 128 KB
    return groupnorm_forward(g, s);
}

// (rows, C, HW) NCHW features of the rollout -> dense NHWC [B][HW][C] (the layout of the compression output), and back.
__global__ void feats_nchw_to_nhwc_kernel(const float* __restrict__ src, const int* __restrict__ rows, float* __restrict__ dst,
                                          int B, int C, int HW) {
    const long long total = (long long)B * C * HW;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const long long t = e / C;
        const int p = (int)(t % HW), f = (int)(t / HW);
        dst[e] = src[((size_t)(rows ? rows[f] : f) * C + c) * HW + p];
    }
}
__global__ void feats_nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int HW) {
    const long long total = (long long)B * C * HW;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(e % HW);
        const long long t = e / HW;
        const int c = (int)(t % C), f = (int)(t / C);
        dst[e] = src[((size_t)f * HW + p) * C + c];
    }
}

static int resnet_backbone_forward(hab_policy* e, const hab_obs* obs, const int* rows, int B, hipStream_t s);

int resnet_feature_shape(const hab_policy* e, int* c, int* hf, int* wf) {
    const ResNetPlan* r = e->rn;
    if (r->blind) return HAB_ERR_UNSUPPORTED;  // a blind encoder has no output (resnet_policy.py:255-257 returns None)
    *c = r->comp_c; *hf = r->comp_fh; *wf = r->comp_fw;
    return HAB_OK;
}

// ResNetEncoder.forward alone: obs -> (n, C, Hf, Wf) NCHW (hab_policy_encode)
int resnet_encode(hab_policy* e, const hab_obs* obs, int n, float* out, hipStream_t s) {
    ResNetPlan* r = e->rn;
    if (r->blind) return HAB_ERR_UNSUPPORTED;
    HAB_TRY(resnet_backbone_forward(e, obs, nullptr, n, s));
    const long long total = (long long)n * r->fc_in;
    feats_nhwc_to_nchw_kernel<<<(int)cdivl(total, 256), 256, 0, s>>>(e->WK + r->comp.w_out, out, n, r->comp_c, r->comp_hw);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

static int resnet_ingest(hab_policy* e, const hab_obs* obs, const int* rows, int B, hipStream_t s);
static int resnet_layers_forward(hab_policy* e, int Btot, int f0, int nB, hipStream_t s);

// Encoder forward of frames [f0, f0 + nB) of a B-frame batch (time-major chunks of a minibatch, engine.hip; f0 = 0, nB = B: the whole
// batch).  The observation ingest and RunningMeanAndVar -- whose batch statistics span ALL B frames (running_mean_and_var.py:33-49) --
// run for the whole batch with the first chunk; everything behind them is per frame.
int resnet_encoder_forward(hab_policy* e, const hab_obs* obs, const uint8_t* masks, const int* rows, int B, hipStream_t s, int f0, int nB) {
    ResNetPlan* r = e->rn;
    float* W = e->WK;
    const int H = e->d.hidden;
    if (nB < 0) nB = B;
    if (f0 < 0 || nB <= 0 || f0 + nB > B || (f0 > 0 && !rows)) return HAB_ERR_ARG;
    if (r->blind) {  // x = the embeddings alone (resnet_policy.py:647-660 skipped, :662-757)
        EmbedArgs ea;
        HAB_TRY(fill_embed_slots(e, obs, ea.slot, false));
        ea.nslots = r->nslots; ea.masks = masks; ea.rows = rows ? rows + f0 : nullptr;
        ea.out = W + e->w_rnnin + (int64_t)f0 * e->rnn_ld; ea.ld = e->rnn_ld; ea.col0 = 0; ea.B = nB;
        ea.saved = W + r->w_embsave + (int64_t)f0 * r->nslots * 4;
        if (!masks) return HAB_ERR_ARG;
        return embed_forward(ea, s);
    }
    if (obs->visual_features) {  // frozen encoder: the rollout already holds its output (resnet_policy.py:636-646)
        if (f0 == 0) {
            const long long total = (long long)B * r->fc_in;
            feats_nchw_to_nhwc_kernel<<<(int)cdivl(total, 256), 256, 0, s>>>(obs->visual_features, rows, W + r->comp.w_out, B, r->comp_c,
                                                                             r->comp_hw);
            HAB_LAUNCH_CHECK();
        }
    } else {
        if (f0 == 0) HAB_TRY(resnet_ingest(e, obs, rows, B, s));
        HAB_TRY(resnet_layers_forward(e, B, f0, nB, s));
    }
    // visual_fc (Flatten in NCHW order -> packed weight is permuted) + ReLU, written into rnn_in[:, :H]
    float* ws = W + e->w_ws;
    float* rin = W + e->w_rnnin + (int64_t)f0 * e->rnn_ld;
    HAB_TRY(linear_fwd(W + r->comp.w_out + (int64_t)f0 * r->fc_in, r->fc_in, e->PK + r->pk_fc, r->fc_in, e->p(r->i_fcb), rin, e->rnn_ld, nB, H,
                       r->fc_in, 1, 0, ws, e->ws_floats, s));
    EmbedArgs ea;
    HAB_TRY(fill_embed_slots(e, obs, ea.slot, false));
    ea.nslots = r->nslots; ea.masks = masks; ea.rows = rows ? rows + f0 : nullptr;
    ea.out = rin; ea.ld = e->rnn_ld; ea.col0 = H; ea.B = nB; ea.saved = W + r->w_embsave + (int64_t)f0 * r->nslots * 4;
    if (!masks) return HAB_ERR_ARG;
    return embed_forward(ea, s);
}

static int resnet_backbone_forward(hab_policy* e, const hab_obs* obs, const int* rows, int B, hipStream_t s) {
    HAB_TRY(resnet_ingest(e, obs, rows, B, s));
    return resnet_layers_forward(e, B, 0, B, s);
}

// observation ingest + RunningMeanAndVar over the WHOLE batch: obs -> x0 [B][H/2][W/2][cpad]
static int resnet_ingest(hab_policy* e, const hab_obs* obs, const int* rows, int B, hipStream_t s) {
    ResNetPlan* r = e->rn;
    float* W = e->WK;
    const hab_policy_desc& d = e->d;
    float* x0 = W + r->w_x0;
    if ((d.has_rgb && !obs->rgb) || (d.has_depth && !obs->depth) || (d.has_semantic && !obs->semantic)) return HAB_ERR_ARG;
    // evaluation mode: RunningMeanAndVar is a fixed affine per channel -> applied by the ingest itself (one launch, one pass less)
    const bool fused_norm = d.normalize_visual_inputs && !e->training;
    const bool moments = d.normalize_visual_inputs && e->training;
    double* ds = reinterpret_cast<double*>(W + r->w_dscratch);  // [0..15] totals, then the per-workgroup partials
    int mom_blocks = 0;
    HAB_TRY(ingest_pool(d.has_rgb ? obs->rgb : nullptr, d.has_depth ? obs->depth : nullptr, d.has_semantic ? obs->semantic : nullptr, rows,
                        x0, B, d.H, d.W, r->cpad, r->c_rgb, r->c_depth, r->c_sem, s, fused_norm ? e->p(r->i_mean) : nullptr,
                        fused_norm ? e->p(r->i_var) : nullptr, moments ? e->p(r->i_mean) : nullptr, moments ? ds + 16 : nullptr, &mom_blocks));
    const long long npix = (long long)B * r->H2 * r->W2;
    r->x0_raw = false;
    if (moments) {
        // training mode (running_mean_and_var.py:24-70): batch moments -> Chan merge into the running ones -> normalise with the UPDATED
        // statistics.  The moments were accumulated by the ingest about the running mean as pivot (resnet_ops.hip); under DD-PPO
        // (:38-49) all_reduce(new_mean), all_reduce(new_count), new_mean /= world; all_reduce(new_var), new_var /= world: the variance is
        // about the CROSS-RANK mean, which the pivot form gives without a second pass.  The callback only SUMS (scale 1); the divisions
        // happen where the sums are consumed, and the count is the real all-reduced number of frames (ranks hold different numbers of
        // frames after a preempted rollout).
        float* st = W + r->w_stats;  // [0..7] batch mean, [8] batch count (frames), [16..23] batch var, [32..47] affine of the new statistics
        const bool dist = e->comm != nullptr || (e->allreduce_cb && e->world_size > 1);  // (a 1-rank communicator still runs: identity)
        auto sum_ranks = [&](float* buf, int n) -> int {  // device-side on the compute stream (comm.hip), else the host callback
            if (e->comm) return hab_comm_allreduce_sum(e->comm, buf, n, s);
            e->allreduce_cb(buf, n, 1.0f, e->allreduce_ctx);
            return HAB_OK;
        };
        const float div = dist ? (float)e->world_size : 1.f;
        HAB_TRY(moment_finish_mean(ds + 16, mom_blocks, r->cpad, e->p(r->i_mean), r->creal, npix, ds, st, st + 8, (float)B, s));
        if (dist) HAB_TRY(sum_ranks(st, 9));
        HAB_TRY(moment_finish_var(ds, e->p(r->i_mean), r->creal, r->cpad, st, div, npix, st + 16, s));
        if (dist) HAB_TRY(sum_ranks(st + 16, 8));
        HAB_TRY(rmv_update(e->p(r->i_mean), e->p(r->i_var), e->p(r->i_count), st, st + 16, (float)B, r->creal, s, st + 8, div, st + 32));
        if (r->stem_takes_raw) r->x0_raw = true;  // the stem kernels normalise while staging
        else HAB_TRY(rmv_normalize(x0, npix, r->cpad, r->creal, e->p(r->i_mean), e->p(r->i_var), s));
    }
    return HAB_OK;
}

// stem, blocks, compression on frames [f0, f0 + B) of a Btot-frame batch: x0 -> comp.w_out [frame][Hf*Wf][C].  Every workspace buffer
// is [frame][...], so a chunk works at frame offset f0 of each; decisions that the backward (which sees the whole batch) must share --
// the fused stem -- are taken for Btot.
static int resnet_layers_forward(hab_policy* e, int Btot, int f0, int B, hipStream_t s) {
    ResNetPlan* r = e->rn;
    float* W = e->WK;
    const int64_t F0 = f0;
    const RnConv& st = r->stem;
    const int64_t stem_out = st.out_floats(), pool_floats = (int64_t)r->poolH * r->poolW * st.cd.Cout;
    float* x0 = W + r->w_x0 + F0 * r->H2 * r->W2 * r->cpad;
    float* stem_raw = W + st.w_raw + F0 * stem_out;
    float* stem_act = W + st.w_out + F0 * stem_out;
    float* pool = W + r->w_pool + F0 * pool_floats;
    uint8_t* pool_idx = reinterpret_cast<uint8_t*>(W + r->w_pool_idx) + F0 * pool_floats;  // one byte per pooled element
    ConvDesc scd = st.cd;
    scd.B = B;
    // the strip kernel leaves the GroupNorm partial statistics of its strips (8 output rows) with the output: no statistics pass
    const int stat_chunks = (scd.Ho() + STEM_STAT_ROWS - 1) / STEM_STAT_ROWS;
    float* gn_part = W + e->w_ws;
    bool have_part = false;
    auto stem_conv = [&](bool want_part) -> int {
        want_part = want_part && (st.groups == 8 || st.groups == 16 || st.groups == 32) &&
                    (size_t)B * stat_chunks * st.groups * 2 <= e->ws_floats;
        const int rcs = st.pk_p >= 0 ? stem_conv_forward(x0, reinterpret_cast<const unsigned short*>(e->PK + st.pk_p), stem_raw, B,
                                                                         scd.H, scd.W, s, r->x0_raw ? W + r->w_stats + 32 : nullptr,
                                                                         want_part ? gn_part : nullptr, st.groups) : 1;
        have_part = want_part && rcs == 0;
        if (rcs != 0 && rcs != 1) return rcs;
        if (rcs == 1 && r->x0_raw) return HAB_ERR_UNSUPPORTED;  // (stem_takes_raw was decided on the same coverage predicates)
        if (rcs == 1) HAB_TRY(conv_fwd(scd, x0, e->PK + st.pk_f, nullptr, stem_raw, 0, W + e->w_ws, e->ws_floats, s));
        return HAB_OK;
    };
    GnArgs g;
    g.x = stem_raw; g.y = nullptr; g.gamma = e->p(st.i_gamma); g.beta = e->p(st.i_beta); g.residual = nullptr;
    g.mean = nullptr; g.rstd = nullptr; g.B = B; g.HW = scd.Ho() * scd.Wo(); g.C = scd.Cout; g.groups = st.groups; g.relu = 1;
    g.eps = 1e-5f; g.scratch = W + e->w_ws; g.scratch_floats = e->ws_floats;
    bool stem_done = false;
    if (!e->save_acts) {  // act / encode: GroupNorm + ReLU + max-pool in one pass, the normalised frame is never written
        HAB_TRY(stem_conv(groupnorm_pool_fusable(B, g.HW, g.C, g.groups, e->ws_floats)));
        const int rc = groupnorm_relu_maxpool_forward(g, scd.Ho(), scd.Wo(), pool, nullptr, s, have_part ? gn_part : nullptr, STEM_STAT_ROWS);
        if (rc != 0 && rc != 1) return rc;
        if (rc == 1) {  // small frames: the register-resident GroupNorm, then the pool
            g.y = stem_act; g.mean = W + st.w_mean + F0 * st.groups; g.rstd = W + st.w_rstd + F0 * st.groups;
            HAB_TRY(groupnorm_forward(g, s));
            HAB_TRY(maxpool_forward(stem_act, pool, pool_idx, B, scd.Ho(), scd.Wo(), scd.Cout, s));
        }
        stem_done = true;
    }
    // (the fused form needs its chunk-parallel kernels for the chunk AND for the whole batch the backward runs on)
    if (!stem_done && groupnorm_pool_fusable(B, g.HW, g.C, g.groups, e->ws_floats) &&
        groupnorm_pool_fusable(Btot, g.HW, g.C, g.groups, e->ws_floats)) {
        // training forward: the same fused pass, keeping the statistics and the arg-max bytes; the ReLU mask is recomputed in the backward
        HAB_TRY(stem_conv(true));
        g.mean = W + st.w_mean + F0 * st.groups; g.rstd = W + st.w_rstd + F0 * st.groups;
        HAB_TRY(groupnorm_relu_maxpool_forward(g, scd.Ho(), scd.Wo(), pool, pool_idx, s, have_part ? gn_part : nullptr, STEM_STAT_ROWS));
        r->stem_fused = true; r->stem_act_valid = false;
        stem_done = true;
    }
    if (!stem_done) {
        HAB_TRY(conv_gn_forward(e, st, x0, nullptr, 1, B, s, f0));
        HAB_TRY(maxpool_forward(stem_act, pool, pool_idx, B, scd.Ho(), scd.Wo(), scd.Cout, s));
        r->stem_fused = false; r->stem_act_valid = true;
    }
    int64_t in_floats = pool_floats;  // per-frame size of the current block's input
    for (const auto& blk : r->blocks) {
        const float* in = W + blk.w_in + F0 * in_floats;
        const float* residual = in;
        if (blk.ds >= 0) {
            const RnConv& dc = r->convs[blk.ds];
            HAB_TRY(conv_gn_forward(e, dc, in, nullptr, 0, B, s, f0));
            residual = W + dc.w_out + F0 * dc.out_floats();
        }
        const float* cur = in;
        for (size_t q = 0; q < blk.convs.size(); ++q) {
            const RnConv& c = r->convs[blk.convs[q]];
            const bool last = q + 1 == blk.convs.size();
            if (last && blk.se_c) HAB_TRY(conv_gn_forward(e, c, cur, nullptr, 0, B, s, f0));  // un-gated: the SE gate comes next
            else HAB_TRY(conv_gn_forward(e, c, cur, last ? residual : nullptr, 1, B, s, f0));
            cur = W + c.w_out + F0 * c.out_floats();
        }
        const RnConv& lc = r->convs[blk.convs.back()];
        if (blk.se_c) {  // out = relu(sigmoid(W2 relu(W1 mean_hw(y) + b1) + b2) * y + identity)
            const int HW = lc.cd.Ho() * lc.cd.Wo(), C = blk.se_c, R = blk.se_r;
            float* ws = W + e->w_ws;
            float* y = W + lc.w_out + F0 * lc.out_floats();
            float* sp = W + blk.w_se_pool + F0 * C; float* sh = W + blk.w_se_h + F0 * R; float* sg = W + blk.w_se_gate + F0 * C;
            HAB_TRY(se_pool(y, sp, B, HW, C, s));
            HAB_TRY(linear_fwd(sp, C, e->p(blk.i_se1w), C, e->p(blk.i_se1b), sh, R, B, R, C, 1, 0, ws, e->ws_floats, s));
            HAB_TRY(linear_fwd(sh, R, e->p(blk.i_se2w), R, e->p(blk.i_se2b), sg, C, B, C, R, 0, 0, ws, e->ws_floats, s));
            HAB_TRY(sigmoid_inplace(sg, (long long)B * C, s));
            HAB_TRY(se_apply_forward(y, sg, residual, W + blk.w_out + F0 * lc.out_floats(), B, HW, C, s));
        }
        in_floats = lc.out_floats();
    }
    return conv_gn_forward(e, r->comp, W + r->blocks.back().w_out + F0 * in_floats, nullptr, 1, B, s, f0);
}

namespace {
struct GPool {  // tiny allocator over the gradient scratch buffers
    float* buf[6];
    bool used[6] = {false, false, false, false, false, false};
    float* get() {
        for (int i = 0; i < 6; ++i)
            if (!used[i]) { used[i] = true; return buf[i]; }
        return nullptr;
    }
    void put(const float* p) {
        for (int i = 0; i < 6; ++i)
            if (buf[i] == p) used[i] = false;
    }
};
}  // namespace

// GroupNorm backward of conv c: dy (optionally masked by relu_out) -> d_raw; gamma / beta gradients.
static int gn_backward(hab_policy* e, const RnConv& c, const float* dy, const float* relu_out, float* d_raw, float* dy_masked, int B,
                       hipStream_t s) {
    float* W = e->WK;
    ResNetPlan* r = e->rn;
    GnBwdArgs g;
    g.x = W + c.w_raw; g.dy = dy; g.relu_out = relu_out; g.dx = d_raw; g.dy_masked = dy_masked; g.gamma = e->p(c.i_gamma);
    g.mean = W + c.w_mean; g.rstd = W + c.w_rstd; g.chan_sums = W + r->w_chansums; g.B = B; g.HW = c.cd.Ho() * c.cd.Wo();
    g.C = c.cd.Cout; g.groups = c.groups;
    g.scratch = W + e->w_ws; g.scratch_floats = e->ws_floats;
    HAB_TRY(groupnorm_backward(g, s));
    const int C = c.cd.Cout;
    HAB_TRY(colsum(W + r->w_chansums, 2 * C, B, C, e->g(c.i_beta), 0, W + e->w_ws, e->ws_floats, s));
    HAB_TRY(colsum(W + r->w_chansums + C, 2 * C, B, C, e->g(c.i_gamma), 0, W + e->w_ws, e->ws_floats, s));
    return HAB_OK;
}

// Weight gradient of one convolution of the backbone; the call sites that land on the generic implicit-GEMM kernel (no strip-resident
// form: 3x3 with >= 128 channels on a side, strided, grouped-as-dense) are bracketed by HAB_PROBE_RN_WGRAD_IM2COL with their algorithmic
// work: FLOPs = 2 x (B Ho Wo) x Cout x (KH KW C), bytes = x + dY read once, dW written once.
static int rn_conv_wgrad(hab_policy* e, const ConvDesc& c, const float* x, const float* dy, float* dw, float* ws, hipStream_t s) {
    const bool im2col = c.KH == 3 && (c.stride > 1 || c.C >= 128 || c.Cout >= 128);
    if (!im2col) return conv_wgrad(c, x, dy, dw, nullptr, ws, e->ws_floats, s);
    const double pix = (double)c.B * c.Ho() * c.Wo();
    Probe pr(e, HAB_PROBE_RN_WGRAD_IM2COL, s, 2.0 * pix * c.Cout * c.KH * c.KW * c.C,
             4.0 * ((double)c.B * c.H * c.W * c.C + pix * c.Cout + (double)c.Cout * c.KH * c.KW * c.C));
    return conv_wgrad(c, x, dy, dw, nullptr, ws, e->ws_floats, s);
}

int resnet_encoder_backward(hab_policy* e, const hab_obs* obs, const uint8_t* masks, const int* rows, int B, hipStream_t s) {
    ResNetPlan* r = e->rn;
    float* W = e->WK;
    float* ws = W + e->w_ws;
    const int H = e->d.hidden;
    const float* dfc = W + e->w_drnnin;  // [B][rnn_ld]; first H columns already carry visual_fc's ReLU mask
    EmbedBwdArgs eb;
    HAB_TRY(fill_embed_slots(e, obs, eb.slot, true));
    eb.nslots = r->nslots; eb.saved = W + r->w_embsave; eb.dout = dfc; eb.ld = e->rnn_ld; eb.col0 = r->blind ? 0 : H; eb.B = B;
    HAB_TRY(embed_backward(eb, ws, e->ws_floats, s));
    if (r->blind) {  // embeddings, recurrent encoder and heads are all there is: the whole arena is final
        grad_tail_ready(e, 0);
        return HAB_OK;
    }
    GPool gp;
    for (int i = 0; i < 6; ++i) gp.buf[i] = W + r->w_gbuf[i];
    // visual_fc
    HAB_TRY(linear_wgrad(dfc, e->rnn_ld, W + r->comp.w_out, r->fc_in, e->g(r->i_fcw), r->fc_in, B, H, r->fc_in, r->comp_c, r->comp_hw,
                         0, ws, e->ws_floats, s));
    HAB_TRY(colsum(dfc, e->rnn_ld, B, H, e->g(r->i_fcb), 0, ws, e->ws_floats, s));
    grad_tail_ready(e, r->i_fcw);  // visual_fc, recurrent encoder, heads are final (the embeddings in front of the encoder are too)
    if (obs->visual_features) return HAB_OK;  // frozen encoder: its parameters get no gradient (the arena slots stay zero)
    float* d_comp = gp.get();  // gradient wrt compression output, masked by its ReLU
    HAB_TRY(linear_dgrad(dfc, e->rnn_ld, e->PK + r->pk_fc, r->fc_in, W + r->comp.w_out, r->fc_in, r->fc_in, d_comp, r->fc_in, B,
                         r->fc_in, H, 0, ws, e->ws_floats, s));
    // compression conv + GN(1)
    float* d_raw = gp.get();
    HAB_TRY(gn_backward(e, r->comp, d_comp, nullptr, d_raw, nullptr, B, s));
    gp.put(d_comp);
    ConvDesc cd = r->comp.cd;
    cd.B = B;
    const float* comp_in = W + r->blocks.back().w_out;
    HAB_TRY(rn_conv_wgrad(e, cd, comp_in, d_raw, e->g(r->comp.i_w), ws, s));
    float* d_out = gp.get();  // gradient wrt the last block's output (pre ReLU mask)
    HAB_TRY(conv_dgrad(cd, d_raw, e->PK + r->comp.pk_d, nullptr, nullptr, d_out, ws, e->ws_floats, s));
    gp.put(d_raw);
    // blocks, last to first
    for (int bi = (int)r->blocks.size() - 1; bi >= 0; --bi) {
        const RnBlock& blk = r->blocks[bi];
        const int n = (int)blk.convs.size();
        const float* in = W + blk.w_in;
        const float* out = W + blk.w_out;
        float* d_pre = gp.get();
        float* cur = gp.get();
        if (!d_pre || !cur) return HAB_ERR_ARG;
        if (blk.se_c) {
            // through the gate: dm = d_out * (out > 0) (= the identity branch's gradient), dgate = sum_hw dm * y, then the two tiny
            // linears backwards, and dy = dm * gate + dpool / HW is what the last GroupNorm receives
            const RnConv& c = r->convs[blk.convs[n - 1]];
            const int HW = c.cd.Ho() * c.cd.Wo(), C = blk.se_c, R = blk.se_r;
            float* scr = W + r->w_se_scr;
            float* dgate = scr; float* dz = scr + (size_t)B * r->cmax; float* dpool = scr + (size_t)2 * B * r->cmax;
            float* dh = scr + (size_t)3 * B * r->cmax;
            HAB_TRY(se_backward_reduce(d_out, out, W + c.w_out, d_pre, dgate, B, HW, C, s));
            HAB_TRY(sigmoid_grad(W + blk.w_se_gate, dgate, dz, (long long)B * C, s));
            HAB_TRY(linear_wgrad(dz, C, W + blk.w_se_h, R, e->g(blk.i_se2w), R, B, C, R, 0, 0, 0, ws, e->ws_floats, s));
            HAB_TRY(colsum(dz, C, B, C, e->g(blk.i_se2b), 0, ws, e->ws_floats, s));
            HAB_TRY(linear_dgrad(dz, C, e->p(blk.i_se2w), R, W + blk.w_se_h, R, R, dh, R, B, R, C, 0, ws, e->ws_floats, s));
            HAB_TRY(linear_wgrad(dh, R, W + blk.w_se_pool, C, e->g(blk.i_se1w), C, B, R, C, 0, 0, 0, ws, e->ws_floats, s));
            HAB_TRY(colsum(dh, R, B, R, e->g(blk.i_se1b), 0, ws, e->ws_floats, s));
            HAB_TRY(linear_dgrad(dh, R, e->p(blk.i_se1w), C, nullptr, 0, 0, dpool, C, B, C, R, 0, ws, e->ws_floats, s));
            float* dy_gn = gp.get();
            if (!dy_gn) return HAB_ERR_ARG;
            HAB_TRY(se_backward_apply(d_pre, W + blk.w_se_gate, dpool, dy_gn, B, HW, C, s));
            HAB_TRY(gn_backward(e, c, dy_gn, nullptr, cur, nullptr, B, s));
            gp.put(dy_gn);
        } else {
            HAB_TRY(gn_backward(e, r->convs[blk.convs[n - 1]], d_out, out, cur, d_pre, B, s));
        }
        gp.put(d_out);
        for (int q = n - 1; q >= 1; --q) {
            const RnConv& c = r->convs[blk.convs[q]];
            const RnConv& pc = r->convs[blk.convs[q - 1]];
            ConvDesc c2 = c.cd;
            c2.B = B;
            if (c.cgroups > 1) {  // dense weight gradient into scratch, block diagonal gathered into the (Cout, C / groups, 3, 3) gradient
                HAB_TRY(rn_conv_wgrad(e, c2, W + pc.w_out, cur, W + r->w_gdense, ws, s));
                HAB_TRY(gather_grouped_wgrad(W + r->w_gdense, e->g(c.i_w), c.cd.Cout, c.cd.C, c.cgroups, c.cd.KH, c.cd.KW, s));
            } else {
                HAB_TRY(rn_conv_wgrad(e, c2, W + pc.w_out, cur, e->g(c.i_w), ws, s));
            }
            float* tmp = gp.get();
            if (!tmp) return HAB_ERR_ARG;
            HAB_TRY(conv_dgrad(c2, cur, e->PK + c.pk_d, W + pc.w_out /* ReLU mask */, nullptr, tmp, ws, e->ws_floats, s));
            gp.put(cur);
            float* nxt = gp.get();
            if (!nxt) return HAB_ERR_ARG;
            HAB_TRY(gn_backward(e, pc, tmp, nullptr, nxt, nullptr, B, s));
            gp.put(tmp);
            cur = nxt;
        }
        const float* add_ptr = d_pre;
        if (blk.ds >= 0) {
            const RnConv& dc = r->convs[blk.ds];
            float* t1 = gp.get();
            if (!t1) return HAB_ERR_ARG;
            HAB_TRY(gn_backward(e, dc, d_pre, nullptr, t1, nullptr, B, s));
            ConvDesc c2 = dc.cd;
            c2.B = B;
            HAB_TRY(rn_conv_wgrad(e, c2, in, t1, e->g(dc.i_w), ws, s));
            float* t2 = gp.get();
            if (!t2) return HAB_ERR_ARG;
            HAB_TRY(conv_dgrad(c2, t1, e->PK + dc.pk_d, nullptr, nullptr, t2, ws, e->ws_floats, s));
            gp.put(t1);
            gp.put(d_pre);
            add_ptr = t2;
        }
        const RnConv& c0 = r->convs[blk.convs[0]];
        ConvDesc c2 = c0.cd;
        c2.B = B;
        HAB_TRY(rn_conv_wgrad(e, c2, in, cur, e->g(c0.i_w), ws, s));
        float* d_in = gp.get();
        if (!d_in) return HAB_ERR_ARG;
        HAB_TRY(conv_dgrad(c2, cur, e->PK + c0.pk_d, nullptr, add_ptr, d_in, ws, e->ws_floats, s));
        gp.put(cur);
        gp.put(add_ptr);
        d_out = d_in;
        // DD-PPO buckets: the stages finish last-to-first and layer4 + compression hold ~3/4 of the convolution stack's bytes, so the
        // arena tail is extended when layer4 and when layer3 are done; their all-reduces overlap the backward of the earlier stages
        // (large activations, few parameters).  layer2, layer1 and the stem go with the final message.
        if (bi == r->nlayers[0] + r->nlayers[1] + r->nlayers[2] || bi == r->nlayers[0] + r->nlayers[1])
            grad_tail_ready(e, r->convs[blk.convs[0]].i_w);
    }
    // maxpool + stem
    float* d_n0 = W + r->w_gstem[0];
    float* d_raw0 = W + r->w_gstem[1];
    if (r->stem_fused && !r->stem_act_valid) {
        // fused stem: the max-pool gradient is gathered and the ReLU mask recomputed inside the GroupNorm backward kernels -- the
        // gradient of the (never stored) normalised stem output is not materialised either
        const RnConv& c = r->stem;
        GnBwdArgs g;
        g.x = W + c.w_raw; g.dy = nullptr; g.relu_out = nullptr; g.dx = d_raw0; g.dy_masked = nullptr; g.gamma = e->p(c.i_gamma);
        g.mean = W + c.w_mean; g.rstd = W + c.w_rstd; g.chan_sums = W + r->w_chansums; g.B = B; g.HW = c.cd.Ho() * c.cd.Wo();
        g.C = c.cd.Cout; g.groups = c.groups; g.scratch = W + e->w_ws; g.scratch_floats = e->ws_floats;
        g.pool_dy = d_out; g.pool_idx = reinterpret_cast<const uint8_t*>(W + r->w_pool_idx); g.beta = e->p(c.i_beta);
        g.pH = c.cd.Ho(); g.pW = c.cd.Wo();
        HAB_TRY(groupnorm_backward(g, s));
        const int C = c.cd.Cout;
        HAB_TRY(colsum(W + r->w_chansums, 2 * C, B, C, e->g(c.i_beta), 0, W + e->w_ws, e->ws_floats, s));
        HAB_TRY(colsum(W + r->w_chansums + C, 2 * C, B, C, e->g(c.i_gamma), 0, W + e->w_ws, e->ws_floats, s));
    } else {
        HAB_TRY(maxpool_backward(d_out, reinterpret_cast<const uint8_t*>(W + r->w_pool_idx), d_n0, B, r->stem.cd.Ho(), r->stem.cd.Wo(),
                                 r->stem.cd.Cout, s));
        HAB_TRY(gn_backward(e, r->stem, d_n0, W + r->stem.w_out, d_raw0, nullptr, B, s));
    }
    ConvDesc c0 = r->stem.cd;
    c0.B = B;
    int rcw = 1;
    if (r->cpad == 4 && c0.Cout == 32)  // strip-resident form (stem_wgrad_strip.h); 1: geometry not covered
        rcw = stem_conv_wgrad(W + r->w_x0, d_raw0, e->g(r->stem.i_w), B, c0.H, c0.W, r->creal, ws, e->ws_floats, s,
                              r->x0_raw ? W + r->w_stats + 32 : nullptr);
    if (rcw != 0 && rcw != 1) return rcw;
    if (rcw == 1 && r->x0_raw) return HAB_ERR_UNSUPPORTED;
    if (rcw == 1) HAB_TRY(conv_wgrad(c0, W + r->w_x0, d_raw0, e->g(r->stem.i_w), nullptr, ws, e->ws_floats, s));
    return HAB_OK;
}

int resnet_tap(hab_policy* e, int which, const float** ptr, int64_t* floats) {
    ResNetPlan* r = e->rn;
    if (r->blind) return HAB_ERR_ARG;  // no visual activations exist
    float* W = e->WK;
    const int64_t B = e->last_B;
    switch (which) {
        case HAB_TAP_ENC_IN:
            if (r->x0_raw) {  // the stem kernels normalised while staging: do it in place now (the backward then reads it as it is)
                if (hipDeviceSynchronize() != hipSuccess) return HAB_ERR_ARG;
                HAB_TRY(rmv_normalize(W + r->w_x0, B * r->H2 * r->W2, r->cpad, r->creal, e->p(r->i_mean), e->p(r->i_var), nullptr));
                if (hipDeviceSynchronize() != hipSuccess) return HAB_ERR_ARG;
                r->x0_raw = false;
            }
            *ptr = W + r->w_x0; *floats = B * r->H2 * r->W2 * r->cpad; return HAB_OK;
        case HAB_TAP_STEM:
            if (r->stem_fused && !r->stem_act_valid) {  // the fused forward skipped it: build it from the kept GroupNorm input + statistics
                const RnConv& c = r->stem;
                GnArgs g;
                g.x = W + c.w_raw; g.y = W + c.w_out; g.gamma = e->p(c.i_gamma); g.beta = e->p(c.i_beta); g.residual = nullptr;
                g.mean = W + c.w_mean; g.rstd = W + c.w_rstd; g.B = (int)B; g.HW = c.cd.Ho() * c.cd.Wo(); g.C = c.cd.Cout; g.groups = c.groups;
                g.relu = 1; g.eps = 1e-5f;
                if (hipDeviceSynchronize() != hipSuccess) return HAB_ERR_ARG;
                HAB_TRY(groupnorm_relu_materialize(g, nullptr));
                if (hipDeviceSynchronize() != hipSuccess) return HAB_ERR_ARG;
                r->stem_act_valid = true;  // the backward now reads it as its ReLU mask (unfused form)
            }
            *ptr = W + r->stem.w_out; *floats = B * r->stem.out_floats(); return HAB_OK;
        case HAB_TAP_POOL: *ptr = W + r->w_pool; *floats = B * r->poolH * r->poolW * r->stem.cd.Cout; return HAB_OK;
        case HAB_TAP_COMPRESSION: *ptr = W + r->comp.w_out; *floats = B * r->comp.out_floats(); return HAB_OK;
        case HAB_TAP_POOL_IDX: *ptr = W + r->w_pool_idx; *floats = (B * r->poolH * r->poolW * r->stem.cd.Cout + 3) / 4; return HAB_OK;
        default: break;
    }
    if (which >= HAB_TAP_CONV_OUT && which < HAB_TAP_CONV_OUT + (int)r->convs.size()) {
        const RnConv& c = r->convs[which - HAB_TAP_CONV_OUT];
        *ptr = W + c.w_out;
        *floats = B * c.out_floats();
        return HAB_OK;
    }
    if (which >= HAB_TAP_LAYER1 && which < HAB_TAP_LAYER1 + 4) {
        // last block of stage (which - HAB_TAP_LAYER1)
        const int* layers = r->nlayers;
        int idx = -1;
        for (int li = 0; li <= which - HAB_TAP_LAYER1; ++li) idx += layers[li];
        const RnBlock& blk = r->blocks[idx];
        *ptr = W + blk.w_out;
        *floats = B * r->convs[blk.convs.back()].out_floats();
        return HAB_OK;
    }
    return HAB_ERR_ARG;
}
