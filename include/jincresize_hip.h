/*
 * jincresize_hip.h -- C ABI of libjincresize_hip.so, the MI355X (gfx950) replacement for the
 * per-frame hot path of Asd-g/AviSynth-JincResize v2.1.4.
 *
 * "ref:" citations are /root/reference/src/JincResize.cpp unless another file is named.
 *
 * What this boundary replaces in the reference plugin:
 *   - Create_JincResize's argument handling + table construction          (ref :654-984)
 *   - (d->*d->process_frame)(src, dst, vi) inside JincResize_GetFrame      (ref :615), i.e. the
 *     resize_plane_{c,sse41,avx2,avx512}<T,thr,subsampled> kernels        (ref :536-601 and
 *     resize_plane_sse41.cpp / _avx2.cpp / _avx512.cpp) and their row dispatch (ref :589-599)
 *   - free_JincResize                                                     (ref :632-647)
 * The AviSynth registration glue (avisynth_c_plugin_init, the Jinc36/64/144/256 aliases,
 * _ChromaLocation frame property) stays on the plugin side and calls these entry points; the
 * binding is shown in INTEGRATION.md.
 *
 * Results are those of the reference's opt=0 C++ path (resize_plane_c): bit-exact for 8..16-bit
 * integer planes and for float planes (strict sequential un-fused fp32 accumulation).
 *
 * Plain pointers and sizes only; no C++ or framework types cross this boundary.  Every function
 * returning int returns 0 on success and a negative jinc_status on failure; the message is
 * available from jinc_last_error() (thread-local) and, for jinc_filter_create, also copied to
 * the caller's buffer because AviSynth reports Create-time errors as strings (ref :682-687).
 */
#ifndef JINCRESIZE_HIP_H
#define JINCRESIZE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define JINC_API __attribute__((visibility("default")))
#else
#define JINC_API
#endif

typedef enum jinc_status {
    JINC_OK = 0,
    JINC_ERR_INVALID_ARG = -1, /* argument rejected with one of the reference's "JincResize: ..." messages */
    JINC_ERR_NO_DEVICE = -2,   /* no usable HIP device / HIP runtime error */
    JINC_ERR_HIP = -3,
    JINC_ERR_NOMEM = -4,
    JINC_ERR_UNSUPPORTED = -5  /* geometry the reference itself handles with undefined behaviour */
} jinc_status;

/* The AVS_VideoInfo facts Create_JincResize / resize_plane_c read from the input clip:
 * avs_is_planar (ref :700), width/height (ref :783-784), avs_bits_per_component (ref :793),
 * avs_num_components (ref :798), avs_is_444 / avs_is_rgb (ref :826), avs_is_420 (ref :744),
 * avs_get_plane_{width,height}_subsampling(vi, AVS_PLANAR_U) (ref :833-834),
 * avs_component_size (ref :903). */
typedef struct jinc_video_info {
    int width;
    int height;
    int bits_per_component; /* 8, 10, 12, 14, 16 or 32 (float) */
    int component_size;     /* bytes per sample: 1, 2 or 4 */
    int num_components;     /* 1 (Y), 3 or 4 (with alpha) */
    int is_planar;          /* non-zero for planar formats */
    int is_rgb;             /* planar RGB(A): plane order G,B,R,A (ref :540) */
    int sub_w;              /* log2 horizontal chroma subsampling (0 for Y/444/RGB) */
    int sub_h;              /* log2 vertical chroma subsampling */
} jinc_video_info;

/* Bits of jinc_args.defined: which optional script arguments were given (avs_defined). */
enum {
    JINC_ARG_SRC_LEFT = 1 << 0,
    JINC_ARG_SRC_TOP = 1 << 1,
    JINC_ARG_SRC_WIDTH = 1 << 2,
    JINC_ARG_SRC_HEIGHT = 1 << 3,
    JINC_ARG_QUANT_X = 1 << 4,
    JINC_ARG_QUANT_Y = 1 << 5,
    JINC_ARG_TAP = 1 << 6,
    JINC_ARG_BLUR = 1 << 7,
    JINC_ARG_CPLACE = 1 << 8,
    JINC_ARG_THREADS = 1 << 9,
    JINC_ARG_OPT = 1 << 10,
    JINC_ARG_INITIAL_CAPACITY = 1 << 11,
    JINC_ARG_INITIAL_FACTOR = 1 << 12
};

/* The script arguments of JincResize(), in registration order (ref :1044-1060), as the plugin
 * reads them in Create_JincResize (ref :703-789).  Arguments whose bit is clear in `defined`
 * take the reference's defaults. */
typedef struct jinc_args {
    int target_width;       /* "i" (required) */
    int target_height;      /* "i" (required) */
    double src_left;        /* [src_left]f   default 0 */
    double src_top;         /* [src_top]f    default 0 */
    double src_width;       /* [src_width]f  default clip width;  <= 0: relative (ref :763-765) */
    double src_height;      /* [src_height]f default clip height; <= 0: relative (ref :768-770) */
    int quant_x;            /* [quant_x]i    default 256, 1..256 */
    int quant_y;            /* [quant_y]i    default 256, 1..256 */
    int tap;                /* [tap]i        default 3, 1..16 */
    double blur;            /* [blur]f       default (and 0) -> 1.0 (ref :772-774) */
    const char *cplace;     /* [cplace]s     "MPEG2" | "MPEG1" | "topleft", case-insensitive */
    int threads;            /* [threads]i    0 or 1 (ref :758-760, :901): 1 keeps the copies of pageable planes on the calling thread, 0 lets large planes use the library's helper threads */
    int opt;                /* [opt]i        -1..3; validated as in the reference, advisory on the GPU path */
    int initial_capacity;   /* [initial_capacity]i > 0; validated, otherwise unused (scratch sizing only) */
    double initial_factor;  /* [initial_factor]f >= 1.0; validated, otherwise unused */
    unsigned defined;       /* JINC_ARG_* bits */
    /* Host facts the reference queries from the script environment: */
    int frame0_chroma_location; /* _ChromaLocation of frame 0 if that property is an int, else -1 = "no such property"
                                   (only consulted when cplace is not given; ref :727-742).  0 / 1 / 2 select the
                                   siting; ANY other integer the property holds -- negative ones too: pass them as 3 --
                                   is the reference's "invalid _ChromaLocation" (switch default, :737) */
    int cpu_has_sse41;      /* avs_get_cpu_flags() & AVS_CPUF_SSE4_1 (ref :755) */
    int cpu_has_avx2;       /* ... & AVS_CPUF_AVX2    (ref :753) */
    int cpu_has_avx512f;    /* ... & AVS_CPUF_AVX512F (ref :751) */
} jinc_args;

/* Opaque filter instance = the reference's `JincResize` object (JincResize.h:39-57) plus its
 * device-resident plan.  One instance is not re-entrant (AviSynth MT_MULTI_INSTANCE, ref :649-652):
 * create one per host thread; instances share nothing mutable. */
typedef struct jinc_filter jinc_filter;

/* Number of usable HIP devices (0 when there is none); replaces nothing, used to shard frames. */
JINC_API int jinc_device_count(void);

/* Round-robin device index for hosts that create one filter instance per worker thread (AviSynth Prefetch(N),
 * MT_MULTI_INSTANCE, ref :649-652): successive calls return 0, 1, ..., jinc_device_count()-1, 0, ... so that the
 * instances -- and with them the frames, which are independent units -- spread over the GPUs of the node with no
 * data exchanged between devices.  Returns -1 when there is no device. */
JINC_API int jinc_pick_device(void);

/* Message of the last failure on the calling thread ("" if none). */
JINC_API const char *jinc_last_error(void);

/* Create_JincResize (ref :654-984): validates the arguments with the reference's rules and error
 * strings, derives crop/chroma geometry (ref :762-866), builds the LUT and the coefficient plan(s)
 * and uploads them to HIP device `device`.  On failure *out is NULL and the message (e.g.
 * "JincResize: tap must be between 1..16.") is copied to err (if err_len > 0). */
JINC_API int jinc_filter_create(const jinc_video_info *vi, const jinc_args *args, int device,
                                jinc_filter **out, char *err, size_t err_len);

/* free_JincResize (ref :632-647). NULL is allowed. */
JINC_API void jinc_filter_free(jinc_filter *f);

/* The output clip's AVS_VideoInfo: the input's with width/height replaced (ref :791-792). */
JINC_API int jinc_filter_output_info(const jinc_filter *f, jinc_video_info *out_vi);

/* Value JincResize_GetFrame writes to the _ChromaLocation frame property (ref :617-625): **2 for every 4:2:0 / 4:2:2 /
 * 4:1:1 output, whatever the siting**; -1 when the property is not written (4:4:4, Y, RGB).  The reference's source
 * reads as "0 mpeg2, 1 mpeg1, 2 topleft", but it compares the member `d->cplace`, which nothing assigns (`new
 * JincResize()` :676; the siting string is the LOCAL `cplace` declared at :715), so the binary always takes the `else`
 * at :623-624.  A drop-in writes what the binary writes.  (Pixels are not affected: the local string drives the chroma
 * geometry, :838-841.) */
JINC_API int jinc_filter_chroma_location(const jinc_filter *f);

/* Private switch, in the manner of jinc_filter_set_simd_order: JINC_CHROMA_LOCATION_BY_SITING makes
 * jinc_filter_chroma_location return what the reference's source means to write -- 0 mpeg2, 1 mpeg1, 2 topleft, by the
 * cplace argument or frame 0's property -- for hosts that want the property to describe the chroma they get.  A
 * deliberate deviation from the reference binary; off by default. */
#define JINC_CHROMA_LOCATION_AS_REFERENCE 0
#define JINC_CHROMA_LOCATION_BY_SITING 1
JINC_API int jinc_filter_set_chroma_location_mode(jinc_filter *f, int mode);

/* The body of JincResize_GetFrame between avs_new_video_frame_p and avs_prop_set_int, i.e.
 * (d->*d->process_frame)(src, dst, vi) (ref :615), on HOST plane buffers as AviSynth hands them
 * over (avs_get_read_ptr_p / avs_get_write_ptr_p, pitches from avs_get_pitch_p in bytes).
 * Planes are indexed in the reference's processing order (ref :539-541): {Y,U,V,A} or {G,B,R,A}.
 * Synchronous: copies to the device, runs the kernels, copies back, returns when dst is complete. */
JINC_API int jinc_filter_get_frame(jinc_filter *f, const void *const src[4], const int src_pitch[4],
                                   void *const dst[4], const int dst_pitch[4]);

/* ---- Look-ahead pipeline around GetFrame (frame transport) -------------------------
 * A host that knows which frames come next (a plugin that prefetches fi->child frames n+1..n+k, a batch tool)
 * keeps up to `depth` (1..256) frames in flight per instance.  The surface stays per frame, as the reference's
 * GetFrame is (ref :603-630): jinc_filter_submit takes ONE frame (its H2D copy is queued at once) and returns a
 * ticket; jinc_filter_wait blocks until THAT frame's destination planes are complete.  In between the library
 * coalesces: `group` consecutively submitted frames share one strided device buffer and ONE set of kernel
 * launches (the batch kernels -- lanes = frames, wide tiles -- that single-frame calls cannot use), followed by one
 * D2H copy per frame.  A group leaves when it is full, when a wait asks for one of its frames, or on
 * jinc_filter_flush.  group = 0 picks depth / 2 (depth >= 8; else 1): one group computes while the client collects
 * the previous one.  src and dst must stay valid and untouched until the frame's wait returns.  Frames are
 * independent, so neither grouping nor completion order changes results.
 * register_host_buffers: how the library treats the caller's plane buffers --
 *   0 (a new instance's state)  pageable, and only the CPU ever touches them: source rows are copied into a pinned buffer of the
 *      library's own at submit (small frames in groups of four or more: when their group is launched), result rows out of one when the
 *      frame's event has fired (in jinc_filter_wait; frames nobody waits for arrive when their group buffer is reused, on
 *      jinc_filter_set_pipeline and on jinc_filter_free).  The DMA engines move whole planes between those buffers and the
 *      device; the device never maps the caller's pages.  Large planes are copied by up to six threads (a process-wide pool of
 *      helpers, idle otherwise) unless the script said threads = 1.  Costs pinned host memory of the size of the device staging
 *      (frames in flight x frame bytes, at most 4 GiB: larger groups are halved).  C2: 2 570 - 2 900 frames/s at one frame in
 *      flight, 5 280 - 6 000 at eight over four boxes (CPU work: it varies with the host; profiles/round6/host_modes*.log).
 *   3  pageable planes handed to the HIP runtime as they are (hipMemcpy2DAsync on the caller's pointers): the default of rounds
 *      1 - 5.  On this ROCm build the runtime maps the caller's pages into the device behind such a copy and keeps the mapping
 *      for a while; C2 3 879 / 4 283 frames/s.  Full test runs and one measuring script of round 6 ended in GPU memory access
 *      faults on heap addresses inside such copies; the cause was not established (profiles/round6/README.md).
 *   any other value  registered once with hipHostRegister (exactly the plane's bytes) and CACHED by address range, least recently
 *      used out: asynchronous copies, results written by the shader, no cost per frame once a buffer has been seen (C2 4 020 /
 *      5 631 / 6 125 frames/s at 1 / 8 / 128 frames in flight).  For hosts whose frame memory is a pool that STAYS MAPPED: the
 *      caller guarantees that such buffers stay allocated until jinc_filter_free or jinc_filter_set_pipeline(f, depth, 0).  The
 *      runtime consults its table of registered ranges for every host pointer it is handed, so a registration that outlives its
 *      pages makes a later buffer at those addresses travel through a dead mapping (a GPU memory access fault) or be refused
 *      (hipErrorInvalidValue when it starts inside the range and runs past its end) -- and the library cannot see a range that
 *      came back at the same addresses.  The frame memory should also OWN ITS PAGES (allocations of whole pages, as large frame
 *      buffers are): planes carved out of the malloc heap share their first and last page with whatever else lives there, and
 *      every GPU memory access fault the tests of this mode ran into in round 6 was on such a heap address; with the planes in
 *      mappings of their own they did not recur (profiles/round6/README.md).
 *   Planes inside a range the caller pinned itself (jinc_filter_adopt_host_range) travel through that mapping in every mode.
 *   Round 6 also built, measured and withdrew "registered at submit, unregistered when the frame's wait returns" (4 508 C2
 *   frames/s): registration at frame rate.
 * A failed launch is reported by the submit that triggered it and by every wait on a frame of that group.
 * jinc_filter_get_frame == submit + wait (after draining frames still in flight). */
JINC_API int jinc_filter_set_pipeline(jinc_filter *f, int depth, int register_host_buffers);
JINC_API int jinc_filter_set_pipeline_group(jinc_filter *f, int depth, int group, int register_host_buffers);
JINC_API int jinc_filter_submit(jinc_filter *f, const void *const src[4], const int src_pitch[4], void *const dst[4],
                                const int dst_pitch[4], long long *ticket);
JINC_API int jinc_filter_flush(jinc_filter *f); /* launch the frames submitted so far (no more are coming) */
/* For hosts that pin their frame memory themselves (a frame pool allocated with hipHostMalloc, or pinned once with
 * hipHostRegister(..., hipHostRegisterPortable)): tells the instance that [base, base + bytes) is pinned and stays so
 * until jinc_filter_free.  Planes inside such a range travel like planes the instance pinned itself (asynchronous
 * copies, results written by the shader), with no registration cost per frame and whatever register_host_buffers
 * says; the instance never unregisters them. */
JINC_API int jinc_filter_adopt_host_range(jinc_filter *f, void *base, size_t bytes);
/* The counterpart: the caller is about to unpin or free [base, base + bytes).  Waits for the instance's frames in flight
 * and forgets every adopted range that touches it (planes there are pageable again unless adopted anew). */
JINC_API int jinc_filter_release_host_range(jinc_filter *f, void *base, size_t bytes);
JINC_API int jinc_filter_wait(jinc_filter *f, long long ticket);

/* Same computation on DEVICE-resident planes, asynchronously on `hip_stream` (a hipStream_t; NULL is
 * the HIP null stream, i.e. ordered with the caller's default-stream work), for a batch of `nframes` independent frames (frames are the
 * sharding unit; no frame reads another).  Plane i of frame n starts at
 * src[i] + n*src_frame_stride[i] bytes (likewise dst).  Pitches/strides in bytes; sample alignment
 * required.  src and dst must not overlap.  Returns after enqueueing. */
JINC_API int jinc_filter_process_device(jinc_filter *f, const void *const src[4], const int src_pitch[4],
                                        const size_t src_frame_stride[4], void *const dst[4],
                                        const int dst_pitch[4], const size_t dst_frame_stride[4],
                                        int nframes, void *hip_stream);

/* Block until everything enqueued on the filter's own stream (jinc_filter_get_frame) has finished.
 * Work given to jinc_filter_process_device is synchronised by the caller through its stream. */
JINC_API int jinc_filter_sync(jinc_filter *f);

/* ---- Frames of a clip sharded over the HIP devices of the node (BASELINE.json configs[4]) --------
 * Frames are independent units (JincResize_GetFrame touches frame n only, ref :603-630) and the plan is read-only, so
 * the shard needs no exchange between devices: frame n is computed on device jinc_shard_device(n, G) = n mod G, every
 * device holds a replica of the plan (one filter instance) and keeps `streams_per_device` (1..256) frames in flight
 * through the look-ahead pipeline above (frames coalesced into groups of streams_per_device / 2 per launch), driven by
 * one host thread per device.  No collective.
 * ndevices <= 0: all visible devices.  register_host_buffers: 0 pageable, copied through the instances' own pinned buffers; 3 pageable,
 * handed to the runtime; any other value: the planes of a jinc_batch_process call are
 * pinned by one registrar thread per device running ahead of the submissions (exact byte ranges, planes that follow each
 * other merged) and stay pinned until jinc_batch_free: the caller keeps them allocated until then (a caller that re-uses its
 * planes call after call pays once).  The worker and the registrar of device d run on the CPUs of d's NUMA node (sysfs numa_node of the
 * device's PCI function); jinc_batch_set_affinity(b, 0) leaves the threads where the scheduler puts them.
 * jinc_batch_device_cpus: the CPUs found for the batch's device_index-th device (returns their number, 0 if unknown).
 * jinc_batch_process: src_planes / dst_planes hold 4 pointers per frame ([frame][plane], planes in the reference's
 * processing order, unused planes NULL), HOST buffers with the given pitches (bytes); returns when every frame is
 * complete.  Errors: first failure's status, message from jinc_batch_last_error(). */
typedef struct jinc_batch jinc_batch;
JINC_API int jinc_shard_device(int frame, int ndevices);
JINC_API int jinc_batch_create(const jinc_video_info *vi, const jinc_args *args, int ndevices, int streams_per_device,
                               int register_host_buffers, jinc_batch **out, char *err, size_t err_len);
JINC_API int jinc_batch_devices(const jinc_batch *b);
JINC_API int jinc_batch_set_affinity(jinc_batch *b, int on);
JINC_API int jinc_batch_device_cpus(const jinc_batch *b, int device_index, int *cpus, int max_cpus);
JINC_API int jinc_batch_device_of_frame(const jinc_batch *b, int frame);
JINC_API int jinc_batch_process(jinc_batch *b, int nframes, const void *const *src_planes, const int src_pitch[4],
                                void *const *dst_planes, const int dst_pitch[4]);
JINC_API void jinc_batch_free(jinc_batch *b);
JINC_API const char *jinc_batch_last_error(void);

/* ---- Jinc36Resize / Jinc64Resize / Jinc144Resize / Jinc256Resize (ref :986-1040, :1061-1108) ----
 * Builds the argument set the alias forwards through avs_invoke("JincResize", ...): the three
 * positional arguments plus, when defined, src_left/top/width/height, quant_x/y, cplace, threads
 * (ref :1007-1029), plus tap = taps (3, 4, 6 or 8; ref :1037).  Everything else stays undefined. */
JINC_API int jinc_alias_args(int taps, const jinc_args *alias_in, jinc_args *out);

/* Compatibility modes for users who diff against the reference's SIMD output: 1 / 2 / 3 reproduce
 * the summation order of its opt = 1 (SSE4.1: 4 lane-partial sums, multiply + add), opt = 2 (AVX2: 8 partial sums, FMA)
 * and opt = 3 (AVX-512: 16 partial sums, FMA) paths bit for bit -- horizontal-sum tree, cvtps_epi32 + packus saturation
 * (to 65535 / 255, not to the clip's peak) and the lower clamp of float sources included (ref resize_plane_sse41.cpp:41-90,
 * resize_plane_avx2.cpp:45-98, resize_plane_avx512.cpp:45-103).  0 (default) = the opt = 0 result, the parity target.
 * A private switch: the public `opt` argument does not select it.  Slow path (no LDS staging). */
JINC_API int jinc_filter_set_simd_order(jinc_filter *f, int order);

/* Plan introspection, kernel-selection knobs for A/B measurements, test hooks and kernel timing live in
 * jincresize_hip_test.h: they are exported by the same library but are not part of the drop-in boundary. */

#ifdef __cplusplus
}
#endif
#endif /* JINCRESIZE_HIP_H */
