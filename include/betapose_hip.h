/* betapose_hip.h -- C ABI of libbetapose_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-frame inference hot path of sjtuytc/betapose
 * (SURVEY.md section 8b).  Plain C: opaque handles, raw device/host pointers and
 * sizes, int status codes (0 = ok, <0 = error; text via bp_last_error()).  Nothing
 * here throws, calls exit() or keeps hidden global engine state, so one engine per
 * rank/stream can coexist.  All "d_" pointers are device (HIP) pointers; `stream`
 * is a hipStream_t passed as void* (NULL = default stream).  Calls on one handle
 * must be serialised by the caller; different handles may be used from different
 * host threads.
 *
 * What each entry point replaces in the reference (paths under
 * /root/reference/3_6Dpose_estimator):
 *   bp_yolo_create*        Darknet(cfgfile, reso) + load_weights(path)      yolo/darknet.py:217-221,365-432
 *                          (C twin: init(cfg, weights, gpu)                 train_YOLO/src/yolo_v2_class.hpp:49)
 *   bp_yolo_forward        Darknet.forward -> [B, 10647, 5+C]               yolo/darknet.py:319-363,129-169
 *   bp_yolo_forward_select + dynamic_write_results (nms hard-wired off)     yolo/util.py:104-223
 *   bp_kpd_create          InferenNet_fast.__init__ / load_state_dict      KPD/src/main_fast_inference.py:26-40
 *   bp_kpd_forward         InferenNet_fast.forward -> [B,50,80,64]         main_fast_inference.py:42-46, models/FastPose.py:28-35
 *   bp_kpd_forward_argmax  + the arg-max half of getPrediction              KPD/src/utils/eval.py:113-131
 *   bp_crop                crop_from_dets + cropBox + box rescale          dataloader.py:354-364,794-835; KPD/src/utils/img.py:242-262
 *   bp_resize_bicubic      transforms.Resize((416,416), 3) + ToTensor      dataloader.py:94-99,162
 *   bp_pipeline_*          DetectionLoader.update -> DetectionProcessor.update -> main loop
 *                          (dataloader.py:330-401,438-457; betapose_evaluate.py:145-176) fused on device
 *   bp_solve_pnp           pnp (cv2.solvePnP + cv2.Rodrigues)                 utils/utils.py:17-41
 *   bp_solve_pnp_ransac    the commented-out cv2.solvePnPRansac variant       utils/utils.py:32-36
 *   bp_pose_nms            pose_nms                                           pPose_nms.py:24-122
 *   bp_png_*, bp_loader_*  cv2.imread on ImageLoader's thread (PNG frames)   dataloader.py:150-179
 *   bp_upload              the H2D of a frame (img.cuda())                    dataloader.py:339
 *   bp_darknet_*           Detector(cfg, weights, gpu) / Detector::detect    train_YOLO/src/yolo_v2_class.cpp:95-317
 *                          (+ init/detect_image/detect_mat/dispose: include/yolo_v2_class_compat.h)
 *   bp_*_set_precision, bp_*_set_policy, bp_stream_create_masked, bp_probe_placement, bp_conv2d, bp_*_tap_*,
 *   bp_*_profile, bp_*_op_stats: no reference counterpart (tuning, measurement and test hooks)
 *
 * Weight streams.  "YOLO stream" = payload of a Darknet .weights file after its
 * header (train_YOLO/src/parser.c:1148-1174): per [convolutional] block in cfg order
 * {bn.bias, bn.scale, bn.mean, bn.var | conv.bias}, conv.weight[out,in,k,k], fp32.
 * "KPD stream" = the FastPose state_dict flattened the same way: convs in
 * module-definition order (preact.conv1; per bottleneck conv1, conv2, conv3,
 * [se.fc.0.weight, se.fc.0.bias, se.fc.2.weight, se.fc.2.bias, downsample.0];
 * duc1.conv, duc2.conv, conv_out) with {bn.bias, bn.weight, running_mean,
 * running_var} (or conv.bias for conv_out) before each conv.weight.  The Python host
 * unpickles the .pkl; C never parses pickle.
 */
#ifndef BETAPOSE_HIP_H
#define BETAPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bp_yolo bp_yolo;
typedef struct bp_kpd bp_kpd;
typedef struct bp_pipeline bp_pipeline;

/* floats per frame in the pipeline result record:
 *   [0..7]   select: idx (int bits; -1 = no detection), x1,y1,x2,y2 (YOLO-input pixels), obj, cls_conf, cls_idx
 *   [8..15]  pt1.x, pt1.y, pt2.x, pt2.y (crop window, frame pixels), box x1,y1,x2,y2 (frame pixels)
 *   [16..]   50 x (argmax idx (int bits), max, left, right, up, down)                       */
#define BP_RESULT_FLOATS 316
#define BP_KP_FLOATS 6
#define BP_SEL_FLOATS 8

const char* bp_last_error(void);
int bp_version(void);
int bp_device_count(void);                      /* yolo_v2_class.hpp:52 get_device_count */
int bp_device_name(int device, char* out, int cap);

/* ---- detector ---- */
int bp_yolo_create(const char* cfg_path, const char* weights_path, int reso, int max_batch, int device, bp_yolo** out);
int bp_yolo_create_from_memory(const char* cfg_text, const float* stream, size_t n_floats, int reso, int max_batch,
                               int device, bp_yolo** out);
/* second engine over the SAME device filters (own activations/workspace): one per concurrent stream */
int bp_yolo_clone(const bp_yolo* y, bp_yolo** out);
void bp_yolo_destroy(bp_yolo* y);
int bp_yolo_rows(const bp_yolo* y);             /* 10647 at reso 416 */
int bp_yolo_attrs(const bp_yolo* y);            /* 5 + classes */
int bp_yolo_forward(bp_yolo* y, const float* d_img_nchw, int batch, float* d_pred, void* stream);
/* d_pred may be NULL; d_sel: [batch][8] */
int bp_yolo_forward_select(bp_yolo* y, const float* d_img_nchw, int batch, float conf, int num_classes, float* d_pred,
                           float* d_sel, void* stream);
/* dynamic_write_results on an existing prediction tensor (yolo/util.py:104-223, NMS hard-wired off):
 * d_pred [batch][rows][attrs] -> d_sel [batch][8] */
int bp_yolo_select(const float* d_pred, int batch, int rows, int attrs, float conf, int num_classes, float* d_sel,
                   void* stream);
/* test/inspection hooks: intermediate layer outputs (dense NCHW copies) */
int bp_yolo_tap_count(const bp_yolo* y);
int bp_yolo_tap_info(const bp_yolo* y, int i, char* name, int cap, int* C, int* H, int* W);
int bp_yolo_tap_copy(bp_yolo* y, int i, int batch, float* d_out_nchw, void* stream);

/* ---- key-point detector ---- */
int bp_kpd_create(const float* stream, size_t n_floats, int n_classes, int max_batch, int device, bp_kpd** out);
int bp_kpd_clone(const bp_kpd* k, bp_kpd** out);
void bp_kpd_destroy(bp_kpd* k);
int bp_kpd_forward(bp_kpd* k, const float* d_inps_nchw, int batch, float* d_hm, void* stream);
/* d_hm may be NULL; d_kp: [batch][50][6] */
int bp_kpd_forward_argmax(bp_kpd* k, const float* d_inps_nchw, int batch, float* d_hm, float* d_kp, void* stream);
int bp_kpd_tap_count(const bp_kpd* k);
int bp_kpd_tap_info(const bp_kpd* k, int i, char* name, int cap, int* C, int* H, int* W);
int bp_kpd_tap_copy(bp_kpd* k, int i, int batch, float* d_out_nchw, void* stream);

/* launch-policy knobs (tuning / tests): split-K target block count, minimum K-chunks (of 32) per slice, maximum
 * slices, forced tile (-1 auto) */
int bp_yolo_set_policy(bp_yolo* y, int sk_target_blocks, int sk_min_chunks, int sk_max_splits, int force_tile);
int bp_kpd_set_policy(bp_kpd* k, int sk_target_blocks, int sk_min_chunks, int sk_max_splits, int force_tile);
/* what the matrix cores multiply, for every conv with Cin % 32 == 0 (activations, accumulation and outputs are fp32 in
 * all modes; converted filter copies are made on first use and shared by clones; a pipeline re-captures its graph):
 *   0  fp32 MFMA (v_mfma_f32_32x32x2_f32);
 *   1  fp16 operands, one fp16 MFMA per product (BASELINE configs[2]; results carry fp16 rounding, ~1e-3);
 *   2  fp32-accurate on the bf16 pipe: operands split exactly into three bf16 terms, six partial products
 *      (dropped terms <= 2^-23 relative, below the fp32 accumulation rounding);
 *   3  mode 1 with fp16 SKIP CONNECTIONS: a residual (YOLO shortcut, ResNet bottleneck add) is read from the fp16 operand plane
 *      its producer wrote for the next convolution, and tensors that only convolutions and residual adds read are not stored as
 *      fp32 at all -- activations travel as fp16 between layers, accumulation stays fp32.  Same stated tolerances as mode 1
 *      (batched fp16 runs are bound by what the layers write: +8.5 % frames/s at 28 frames per launch). */
int bp_yolo_set_precision(bp_yolo* y, int precision);
int bp_kpd_set_precision(bp_kpd* k, int precision);
/* per-op static description: returns number of ops; fills up to cap entries of (flops, bytes) per image */
int bp_yolo_op_stats(const bp_yolo* y, double* flops, double* bytes, int cap);
int bp_kpd_op_stats(const bp_kpd* k, double* flops, double* bytes, int cap);
/* eager run with hipEvent pairs around every fused-conv kernel (not its split-K reduce) and every other op:
 * ms[i] = mean device time of op i over `iters` runs; info[i*4..] = (is_conv, tile id, vec path, splits).
 * Returns the number of ops (arrays may be NULL to query). */
int bp_yolo_profile(bp_yolo* y, int batch, int iters, float* ms, int* info, int cap, void* stream);
/* in-situ timing of the convolutions WHILE the pipeline runs (several frames in flight, graph replay): every conv launch
 * whose grid has at most `slots` blocks writes 8 u64 marks of the device-wide 100 MHz reference clock (s_memrealtime) per block (entry, index math done, -, K loop done,
 * stores done, slab parked, slices combined, -) at d_buf + (conv ordinal * slots + block) * 8.  NULL switches it off; a
 * captured pipeline graph is rebuilt on the next run.  op_name: the layer behind op i of bp_*_op_stats / bp_*_profile; returns 1 for a convolution, 0 for any other op, -1 on error. */
/* how long `ticks` marks of the clock those stamps read take (one thread spinning on s_memtime between two events) */
int bp_calibrate_ticks(long long ticks, float* ms, void* stream);
/* lone-frame latency mode (off by default): split-K launches keep all K slices of an output tile on ONE XCD and hand the
 * partial sums over inside that XCD's L2 (checked in every launch against the XCC_ID the hardware reports: a block that is
 * not where the round-robin dispatch puts it raises an error word instead of reading stale sums, bp_*_xcd_errors below), and every convolution launch
 * carries blocks that pull the NEXT convolution's filters into the L2 of the XCD that will read them.  +4.7 % frames/s
 * with one frame at a time (fp16 +4.8 %), a loss of 1-2 % with two or more frames in flight (no idle CUs to spare); results
 * are bit-identical either way.  A captured pipeline graph is rebuilt on the next run. */
int bp_yolo_set_prefetch(bp_yolo* y, int on);
int bp_kpd_set_prefetch(bp_kpd* k, int on);
/* conv -> conv fusion (round 5; default on, bf16x3 mode): a Darknet-53 residual block's 1x1 + 3x3 + shortcut (yolo/darknet.py:319-363)
 * and a bottleneck's conv1 + conv2 (+ conv3 + skip connection; KPD/src/models/layers/SE_Resnet.py:25-42) run as ONE launch on 8 x 8 output
 * patches where the block's maps are large and its channel counts small (208x208 / 104x104 detector blocks, 80x64 key-point blocks): the
 * intermediate tensors stay in LDS.  Same results as the unfused plan within fp32 rounding (a different summation order in the 3x3).
 * set_fusion(0) restores one launch per convolution; fused_launches reports how many groups the current plan fuses at `batch`.  A captured
 * pipeline graph is rebuilt on the next run. */
int bp_yolo_set_fusion(bp_yolo* y, int on);
int bp_kpd_set_fusion(bp_kpd* k, int on);
int bp_yolo_fused_launches(bp_yolo* y, int batch, int* launches);
int bp_kpd_fused_launches(bp_kpd* k, int batch, int* launches);
/* the latency mode's placement check: *count != 0 when, since the last call, a split-K launch found a K slice on another XCD
 * than its reducing block.  Such a launch raises an error word and does NOT store the affected tile (no trap: the context and the
 * other streams live on), so the frame's results are invalid: switch the mode off (bp_*_set_prefetch(., 0)) and run the frame
 * again.  Waits for `stream`, clears the word.  Always 0 outside the latency mode. */
int bp_yolo_xcd_errors(bp_yolo* y, int* count, void* stream);
int bp_kpd_xcd_errors(bp_kpd* k, int* count, void* stream);
int bp_yolo_set_stamps(bp_yolo* y, unsigned long long* d_buf, int slots);
int bp_kpd_set_stamps(bp_kpd* k, unsigned long long* d_buf, int slots);
int bp_yolo_op_name(const bp_yolo* y, int i, char* out, int cap);
int bp_kpd_op_name(const bp_kpd* k, int i, char* out, int cap);
int bp_kpd_profile(bp_kpd* k, int batch, int iters, float* ms, int* info, int cap, void* stream);
size_t bp_yolo_device_bytes(const bp_yolo* y);
size_t bp_kpd_device_bytes(const bp_kpd* k);

/* ---- stand-alone device stages ---- */
/* d_frames: [batch][H][W][3] u8 BGR.  Either d_sel ([batch][8], box in reso-pixel units, rescaled by W/reso, H/reso)
 * or d_boxes ([batch][4] frame-pixel x1,y1,x2,y2) gives the boxes.  Outputs: d_out_nchw [batch][3][oh][ow] and/or
 * d_out_nhwc [batch][oh][ow][3]; d_pts [batch][8]. */
int bp_crop(const uint8_t* d_frames, int batch, int H, int W, const float* d_sel, int reso, const float* d_boxes,
            float* d_out_nchw, float* d_out_nhwc, float* d_pts, int oh, int ow, void* stream);
/* Pillow-exact antialiased bicubic: d_in [batch][H][W][3] u8 -> d_out_u8 [batch][oh][ow][3] (nullable) and/or
 * d_out_nhwc f32 /255 (nullable).  swap_rb: read BGR, write RGB. */
int bp_resize_bicubic(const uint8_t* d_in, int batch, int H, int W, int oh, int ow, int swap_rb, uint8_t* d_out_u8,
                      float* d_out_nhwc, void* stream);
/* the arg-max half of getPrediction on an existing heat-map tensor (KPD/src/utils/eval.py:113-131): d_hm [batch][C][H][W]
 * -> d_kp [batch][C][6] = (flat arg-max index as int bits -- first maximum wins --, max, left, right, up, down; the four
 * neighbours are 0 when the maximum lies on the border) */
int bp_heatmap_argmax(const float* d_hm, int batch, int C, int H, int W, float* d_kp, void* stream);
/* one fused convolution on device tensors (unit tests / kernel benchmarks).  h_w: host OIHW filter, h_bias host or NULL.
 * d_in NHWC [N,H,W,Cin]; d_out per store_mode (0 NHWC, 1 nearest-x2 NHWC, 2 PixelShuffle(2) NHWC, 3 NCHW);
 * act 0 linear / 1 leaky(0.1) / 2 relu; d_res NHWC residual or NULL; splits 0 auto; tile -1 auto, else a kernel id
 * (csrc/bp_common.h ConvTile) plus the operand mode: 0 = 64x64 block, 1 = 128x64 (fp32 MFMA); 13..16 = conv_pl.hip (both
 * operands by LDS-DMA from 16-bit planes: 64x64, 128x128, 128x64, 256x128 blocks); + 256 fp16 operands, + 512 bf16x3.
 * bp_conv2d_planes additionally returns the operand planes the epilogue emits for the next layer
 * (d_out_planes [np][output elements] u16: one fp16 plane or three bf16 planes that sum to the fp32 output exactly). */
int bp_conv2d(const float* d_in, int N, int H, int W, int Cin, const float* h_w, const float* h_bias, int Cout, int k,
              int stride, int pad, int act, int store_mode, const float* d_res, int res_after_act, int tile, int splits,
              float* d_out, int iters, float* ms_per_iter, void* stream);
int bp_conv2d_planes(const float* d_in, int N, int H, int W, int Cin, const float* h_w, const float* h_bias, int Cout, int k,
                     int stride, int pad, int act, int store_mode, const float* d_res, int res_after_act, int tile, int splits,
                     float* d_out, unsigned short* d_out_planes, int iters, float* ms_per_iter, void* stream);

/* ---- whole frame on device: resize -> detector -> select -> crop -> KPD -> arg-max, optionally as one hipGraph ---- */
/* d_frames [batch][H][W][3] u8 BGR, d_results [batch][BP_RESULT_FLOATS], d_hm [batch][50][80][64]: caller-owned
 * device buffers (NULL -> allocated and owned by the pipeline). */
int bp_pipeline_create(bp_yolo* y, bp_kpd* k, int frame_h, int frame_w, int batch, float conf, int num_classes,
                       uint8_t* d_frames, float* d_results, float* d_hm, bp_pipeline** out);
void bp_pipeline_destroy(bp_pipeline* p);
uint8_t* bp_pipeline_frames(bp_pipeline* p);    /* device [batch][H][W][3] u8 BGR, caller fills */
int bp_pipeline_kernel_count(bp_pipeline* p);   /* kernel launches per run (after the first graph capture) */
float* bp_pipeline_results(bp_pipeline* p);     /* device [batch][BP_RESULT_FLOATS] */
float* bp_pipeline_heatmaps(bp_pipeline* p);    /* device [batch][50][80][64] */
int bp_pipeline_set_fixed_box(bp_pipeline* p, const float* box_xyxy_or_null);
int bp_pipeline_run(bp_pipeline* p, int use_graph, void* stream);
/* While either engine is in the lone-frame latency mode (bp_*_set_prefetch) bp_pipeline_run waits for the frame, reads the engines'
 * placement error words and, on a fault, clears them, switches the mode off for both engines and runs the same frame again on the
 * ordinary hand-off: every caller gets a valid record.  bp_pipeline_latency_faults = frames re-run that way so far (-1: null). */
int bp_pipeline_latency_faults(const bp_pipeline* p);
/* Set-up step: capture and instantiate the frame's hipGraph now (records the launches, executes nothing), so that the first
 * bp_pipeline_run(use_graph = 1) is a plain graph launch.  Called again after a precision / policy change it rebuilds the graph.
 * (It creates the pipeline's capture stream: call it AFTER the caller's own streams have launched something -- HIP binds streams to
 * its four hardware queues as they are first used, and two pipelines prepared first were seen to share one queue.) */
int bp_pipeline_prepare(bp_pipeline* p);

/* ---- host post-processing (no device work) ---- */
/* pnp (utils/utils.py:17-41): a restatement of cv2.solvePnP's default SOLVEPNP_ITERATIVE (planar / DLT initialisation,
 * CvLevMarq on (Rodrigues vector, t): <= 20 steps, FLT_EPSILON) followed by cv2.Rodrigues; f64.
 * pts3d [n][3], pts2d [n][2], K [9] row-major; outputs R [9] row-major, t [3].  n >= 6 (>= 4 for a planar model). */
int bp_solve_pnp(const double* pts3d, const double* pts2d, int n, const double* K, double* R, double* t);
/* opt-in, NOT what the reference calls: Hartley-conditioned DLT + the same reprojection objective minimised to
 * convergence -- for callers who want the optimum where the raw-DLT start of SOLVEPNP_ITERATIVE lands in a wrong basin
 * (small distant objects).  n >= 6, non-planar. */
int bp_solve_pnp_refined(const double* pts3d, const double* pts2d, int n, const double* K, double* R, double* t);
/* the variant utils/utils.py:32-36 keeps commented out (cv2.solvePnPRansac, reprojectionError = 12): 6-point
 * hypotheses through the solver above, reproducible sampler; inliers [n] (nullable) receives the consensus mask. */
int bp_solve_pnp_ransac(const double* pts3d, const double* pts2d, int n, const double* K, double reproj_err,
                        int max_trials, double confidence, double* R, double* t, unsigned char* inliers);
/* pose_nms (pPose_nms.py:24-122), f32: bboxes [n][4], bbox_scores [n], preds [n][K][2], scores [n][K] -> returns m <= n
 * merged poses (or a negative status): pick [m] (candidate kept), pose [m][K][2] (the - 0.3 applied), score [m][K],
 * proposal score [m].  Output arrays sized for n. */
int bp_pose_nms(const float* bboxes, const float* bbox_scores, const float* preds, const float* scores, int n, int K,
                int* out_pick, float* out_pose, float* out_score, float* out_prop);

/* ---- Darknet-API-compatible detector (replaces the reference's CPU/CUDA Detector, train_YOLO/src/yolo_v2_class.cpp) ----
 * cfg WITH a [net] block (width == height); BatchNorm folded the Darknet-C way; detections as Detector::detect makes
 * them: candidates with objectness > thresh, prob = objectness * class probability, per-class NMS, boxes in image
 * pixels.  The six yolo_v2_class symbols themselves are declared in include/yolo_v2_class_compat.h. */
typedef struct bp_darknet bp_darknet;
typedef struct bp_bbox {
    unsigned int x, y, w, h;
    float prob;
    unsigned int obj_id, track_id, frames_counter;
} bp_bbox;
const char* bp_darknet_last_error(void);
int bp_darknet_create(const char* cfg_path, const char* weights_path, int device, bp_darknet** out);
void bp_darknet_destroy(bp_darknet* d);
int bp_darknet_width(const bp_darknet* d);
int bp_darknet_height(const bp_darknet* d);
int bp_darknet_classes(const bp_darknet* d);
/* planar_rgb: host [3][h][w] floats 0..1 (Darknet's `image`).  Returns the number of detections (the first `cap` are
 * written to out), < 0 on error. */
int bp_darknet_detect_rgb(bp_darknet* d, const float* planar_rgb, int w, int h, float thresh, float nms, bp_bbox* out,
                          int cap);
/* encoded image in memory: PNG, baseline JPEG or uncompressed BMP (what load_image / stb_image hands the reference's
 * Detector, image.c:1820-1875); bp_darknet_detect_png is the older name of the same entry point */
int bp_darknet_detect_image(bp_darknet* d, const unsigned char* data, size_t n, float thresh, float nms, bp_bbox* out,
                            int cap);
int bp_image_decode_rgb(const unsigned char* data, size_t n, unsigned char* out_rgb, size_t cap, int* h, int* w);
int bp_darknet_detect_png(bp_darknet* d, const unsigned char* png, size_t n, float thresh, float nms, bp_bbox* out,
                          int cap);
int bp_darknet_detect_file(bp_darknet* d, const char* png_path, float thresh, float nms, bp_bbox* out, int cap);
/* detector engine from a Darknet cfg/.weights pair with Darknet-C BatchNorm folding (used by bp_darknet_create) */
int bp_yolo_create_darknet(const char* cfg_path, const char* weights_path, int reso, int max_batch, int device,
                           bp_yolo** out);

/* ---- HIP streams confined to a subset of the CUs (one frame pipeline per XCD) ---- */
/* cu_mask: `words` x 32 bits, bit i = CU i of the device in the driver's numbering (on MI355X bit i lies on XCD i % 8,
 * checked by bp_probe_placement).  The stream is a plain hipStream_t (void*) usable with every call above. */
int bp_stream_create_masked(const uint32_t* cu_mask, int words, void** out_stream);
int bp_stream_destroy(void* stream);
/* launches `blocks` workgroups on `stream` and reports where each ran: h_xcc[b] = XCD id (0..7), h_hw_id[b] = raw
 * HW_ID register (may be NULL) */
int bp_probe_placement(int blocks, int* h_xcc, int* h_hw_id, void* stream);

/* ---- frame input (host; replaces cv2.imread on ImageLoader's thread, dataloader.py:150-179) ---- */
/* PNG -> cv2.imread(IMREAD_COLOR) convention: [h][w][3] u8 in B,G,R order; alpha dropped, grey replicated, palette
 * expanded, 16-bit samples reduced to the high byte.  Adam7-interlaced files are rejected. */
int bp_png_info(const unsigned char* data, size_t n, int* h, int* w, int* channels);
int bp_png_decode_bgr(const unsigned char* data, size_t n, unsigned char* out_bgr, size_t cap, int* h, int* w);
/* read-ahead loader: `threads` workers decode the PNG files `paths[0..n)` (each H x W) in list order into a ring of
 * `depth` host slots (pinned with hipHostMalloc when `pinned` and a GPU is present).
 * bp_loader_next: 0 = *bgr points at frame *index until bp_loader_release(*index); 1 = list exhausted;
 * <0 = that frame failed (bp_last_error(); it must still be released).  One consumer thread. */
typedef struct bp_loader bp_loader;
int bp_loader_create(const char* const* paths, int n, int H, int W, int threads, int depth, int pinned, bp_loader** out);
void bp_loader_destroy(bp_loader* l);
int bp_loader_next(bp_loader* l, long long* index, const unsigned char** bgr);
int bp_loader_release(bp_loader* l, long long index);
/* asynchronous host -> device copy on `stream` (h_src should be a loader slot or other pinned memory) */
int bp_upload(void* d_dst, const void* h_src, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BETAPOSE_HIP_H */
