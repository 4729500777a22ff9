/*
 * jincresize_hip_test.h -- introspection, A/B knobs, test hooks and kernel timing of libjincresize_hip.so.
 *
 * Exported by the same library as the C ABI of jincresize_hip.h, but NOT part of the drop-in boundary: nothing here
 * replaces a piece of the reference plugin.  Used by tests/, bench.py and profiles/.
 */
#ifndef JINCRESIZE_HIP_TEST_H
#define JINCRESIZE_HIP_TEST_H

#include "jincresize_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Plan introspection (tests, tools, benchmarks) -------------------------------------------- */
typedef struct jinc_plan_info {
    int src_width, src_height, dst_width, dst_height;
    int filter_size;      /* ref :356 */
    int num_sets;         /* distinct coefficient sets kept (the reference keeps one per border pixel) */
    int periodic;         /* 1 when the interior was recognised as phase-periodic (fast kernel) */
    int period_x, period_y;   /* output-pixel period of the interior phase pattern */
    int step_x, step_y;       /* source-pixel advance per period */
    int interior_x0, interior_x1, interior_y0, interior_y1; /* half-open interior rectangle */
    int64_t plan_bytes;   /* bytes of the device-resident plan for this table */
    int quasi;            /* 1 when the window origins are affine per residue (quasi-periodic kernel applies) */
    int quasi_period_x, quasi_period_y, quasi_step_x, quasi_step_y;
} jinc_plan_info;

/* table: 0 = luma / all planes, 1 = chroma table of subsampled formats (ref :552-558). */
JINC_API int jinc_filter_num_tables(const jinc_filter *f);
JINC_API int jinc_filter_plan_info(const jinc_filter *f, int table, jinc_plan_info *out);
/* Expands the compact plan back to the reference's per-pixel view for output pixel (x, y):
 * start_x/start_y (EWAPixelCoeffMeta, JincResize.h:11-16) and the filter_size^2 coefficients
 * (row-major, no stride padding) that pixel uses.  coeffs may be NULL. */
JINC_API int jinc_filter_plan_pixel(const jinc_filter *f, int table, int x, int y, int *start_x, int *start_y,
                                    float *coeffs);
/* Bulk form: fills start_x[dst_w], start_y[dst_h] and set_id[dst_h*dst_w]; any pointer may be NULL. */
JINC_API int jinc_filter_plan_dump(const jinc_filter *f, int table, int *start_x, int *start_y, int *set_id);
/* Rectangles of a drifting plan (quasi && !periodic): what the runs form of the direct kernel walks (csrc/plan.h PlanRun, 8 ints
 * each: set, x0, y0, sx0, sy0, ni, nj, first_item).  *n_runs / *n_items receive the counts (0 / 0 when the plan has no runs);
 * up to `capacity` rectangles are copied to `runs` when it is not NULL.  Host code: works on instances without a device. */
JINC_API int jinc_filter_plan_runs(const jinc_filter *f, int table, int *n_runs, int *n_items, int32_t *runs, int capacity);
/* Copies the coefficient set `set` (filter_size^2 floats). */
JINC_API int jinc_filter_plan_set(const jinc_filter *f, int table, int set, float *coeffs);
/* The 1024-entry LUT (ref :265-275) as doubles. */
/* batch.cpp's NUMA lookup against a sysfs tree given by the caller (a fake one in tests/test_batch_affinity.py): the CPUs of
 * the NUMA node of PCI function `bdf` under `sysfs_root`; returns their number (0: unknown node, numa_node = -1, ...). */
/* Registrar threads of jinc_batch_process (0: one per device, the default): lets a one-device box run several side by side. */
struct jinc_batch;
JINC_API int jinc_debug_batch_set_registrars(struct jinc_batch *b, int n);
/* Host ranges hipHostRegister refused since the batch was created (their planes travel pageable); `first`: what the first was told. */
JINC_API int jinc_debug_batch_refused(struct jinc_batch *b, char *first, size_t first_len);
/* hipHostRegister minus hipHostUnregister calls of this library that succeeded: the host ranges it holds pinned right now. */
JINC_API long long jinc_debug_host_registrations(void);
JINC_API int jinc_debug_numa_cpus(const char *sysfs_root, const char *bdf, int *cpus, int max_cpus);
JINC_API int jinc_filter_lut(const jinc_filter *f, double *lut1024);

/* Kernel selection override for tests/benchmarks: 0 = automatic, 1 = force the generic gather
 * kernel for every pixel, 2 = periodic fast kernel where the plan allows (same as automatic),
 * 3..6 = A/B variants of the periodic kernels (row-streamed, other tile heights, packed math),
 * 7 = the quasi-periodic kernel wherever it applies (it is the automatic choice only for drifting ratios),
 * 8 = its waterfall variant (coefficient sets in SGPRs, one pass per distinct set of a wave) and 10 = its per-lane
 * coefficient variant (the default for drifting ratios) on every plan they apply to, 9 = the direct (no-LDS) periodic kernel wherever
 * the plan is exactly periodic (it is the automatic choice for down-scales and taps > 8), 11 = the frame-lane kernel
 * (lanes of a wave = frames of the batch; the automatic choice for batches of >= 16 frames whose plan has no phase
 * structure) for every plan and batch size, always in its 64-frame form, 12 = its frame-pair form (two frames per lane,
 * 128 frames per workgroup; the automatic choice for whole groups of 128 frames, filter sizes 5 and 7) for the whole batch
 * wherever it is configured, 13 = the quad form of the periodic kernel (2x up-scales, filter sizes 7 and 9) wherever it is
 * configured, 14 = the runs form of the direct kernel (drifting plans cut into rectangles of one coefficient set each; the
 * automatic choice for drifting plans with filter sizes above 9) wherever the plan has runs, 15 = the automatic choice on the
 * reference's full window: integer planes otherwise run the periodic kernels on the TRIMMED support (the bounding box of
 * the phase sets' non-zero coefficients -- 6 x 6 of 7 x 7 for the 2x up-scale with tap 3; leaving out taps whose coefficient is
 * 0.0f is exact for finite samples), 16 = the frame-lane kernel for every plan and batch size as 11, groups of fewer than 64
 * frames in its sub-group form (a wave = 8 / 16 / 32 frames x 8 / 4 / 2 output rows; filter sizes 5, 7, 8, 9; the automatic
 * choice for what a batch leaves below 64 frames). */
JINC_API int jinc_filter_set_kernel_mode(jinc_filter *f, int mode);
/* The modes above by name. */
typedef enum jinc_kernel_mode {
    JINC_KM_AUTO = 0,
    JINC_KM_GATHER = 1,            /* ewa_gather_kernel for every pixel */
    JINC_KM_PERIODIC = 2,          /* as automatic, quad forms excluded */
    JINC_KM_ROWS = 3,              /* ewa_periodic_rows_kernel for every filter size */
    JINC_KM_WINDOW_HALF_TILES = 4, /* ewa_periodic_kernel on half-height tiles */
    JINC_KM_PACKED_RG4 = 5,        /* ewa_periodic_pk_kernel (fs 7, full window), 4 / 8 row groups per tile */
    JINC_KM_PACKED_RG8 = 6,
    JINC_KM_QUASI = 7,             /* ewa_quasi_kernel wherever it applies */
    JINC_KM_QUASI_WATERFALL = 8,
    JINC_KM_DIRECT = 9,            /* ewa_direct_kernel wherever the plan is exactly periodic */
    JINC_KM_QUASI_LANE = 10,
    JINC_KM_FRAMELANE = 11,        /* 64-frame form for every plan and batch size */
    JINC_KM_FRAMELANE_PAIR = 12,   /* 128-frame form for the whole batch */
    JINC_KM_QUAD = 13,             /* the quad form configured for the plan's (trimmed) support: jinc_filter_last_instance says which */
    JINC_KM_RUNS = 14,             /* runs form of the direct kernel */
    JINC_KM_FULL_WINDOW = 15,      /* automatic choice, no trimmed support */
    JINC_KM_FRAMELANE_SUB = 16     /* frame-lane kernel, groups below 64 frames in the sub-group form */
} jinc_kernel_mode;
/* Full template instantiation of the kernel that computed the interior of `table` in the most recent frame call, spelled as
 * rocprofv3 prints it (e.g. "ewa_periodic_quad2_kernel<unsigned char, 8, 1026u, 6>": sample type, row groups per tile, chord
 * pattern, taps per kernel row), for the kernel families whose launchers choose between instantiations by call size; the plain
 * kernel name for the others.  bench.py reports it as roofline.kernel and the parity tests assert it, so that the
 * instantiation a benchmark times is one the parity tests have checked. */
JINC_API const char *jinc_filter_last_instance(const jinc_filter *f, int table);

/* ---- A/B and tuning knobs ---------------------------------------------------------------------
 * Process-wide; every knob is UNSET by default and the library never reads the environment for them (bench.py translates the
 * JINC_* variables the profiles/ scripts pass into calls of jinc_debug_set_knob).  Knobs that shape a device plan (TRIM,
 * QUAD_INNER, FL_*, FLP_*, QUASI_LDS_KB) are read by jinc_filter_create; the others by the frame call. */
typedef enum jinc_knob {
    JINC_KNOB_TRIM = 0,               /* 0: no trimmed support (periodic and direct kernels keep the full window) */
    JINC_KNOB_QUAD_INNER,             /* 0: no chord-row patterns in the quad forms */
    JINC_KNOB_RUNS_FL_BORDER_FRAMES,  /* frames per call from which a runs-form plan's border goes to the frame-lane kernel (0: never) */
    JINC_KNOB_PLANE_FORK,             /* 0: planes of small calls stay on one stream */
    JINC_KNOB_PLANE_PAIR,             /* 0: U and V of a single frame as two launches */
    JINC_KNOB_QUASI_SPLIT,            /* workgroups per tile of the quasi-periodic kernel */
    JINC_KNOB_FL_SUB,                 /* sub-groups per wave of the frame-lane kernel's sub-group form (0: never) */
    JINC_KNOB_FLOAT_TRIM_MIN_TAPS,    /* taps per plane and call from which float planes take the trimmed support */
    JINC_KNOB_FLOAT_TRIM_MIN_FS,
    JINC_KNOB_QUAD8,                  /* 0: window kernel, 1: quad form on the 8 x 8 support */
    JINC_KNOB_FL_COLS_FRAMES,         /* frames per call from which border columns go to the frame-lane kernel (0: never) */
    JINC_KNOB_QUAD_RG,                /* 8 / 4: full / half-height tiles of the quad forms whatever the call size */
    JINC_KNOB_QUAD2X8,                /* 0 / 1: two periods per lane on the 8 x 8 support */
    JINC_KNOB_FLOAT_SCAN,             /* 1: finite-sample scan pass in front of the trimmed launch (round 4's first form) */
    JINC_KNOB_FL_FILL_WEIGHT,
    JINC_KNOB_FL_VARIANT,             /* 1: row-segment form of the frame-lane kernel always */
    JINC_KNOB_FL_1K,
    JINC_KNOB_FL_LDS_KB,
    JINC_KNOB_FL_COLW,
    JINC_KNOB_FL_THREADS,
    JINC_KNOB_FLP_LDS_KB,
    JINC_KNOB_FLP_COLW,
    JINC_KNOB_FLP_THREADS,
    JINC_KNOB_BLIT_WORKGROUPS,
    JINC_KNOB_GROUP_SHARES,
    JINC_KNOB_PIPELINE_SKIP,          /* diagnosis only (wrong results): 1 = no H2D copies, 2 = no kernels */
    JINC_KNOB_PIPELINE_DMA,           /* 1: results leave by DMA copies even when every plane is pinned */
    JINC_KNOB_D2H_PRIORITY,           /* departures stream: 0 lowest, 1 highest (default), 2 normal priority */
    JINC_KNOB_QUASI_LDS_KB,
    JINC_KNOB_BLIT_SETPRIO,
    JINC_KNOB_DIRECT_SHAPE,           /* as jinc_debug_set_direct_shape */
    JINC_KNOB_GATHER_PASSES,
    JINC_KNOB_ROWS_PAIR,              /* 0: rows kernel instead of its packed phase-pair form (round 5); 64 / 32 / 16: that tile shape */
    JINC_KNOB_ROWPAIR_SMALL,          /* 1: ewa_periodic_rowpair_kernel also on 6 .. 9 taps per kernel row (default: the window / quad forms there) */
    JINC_KNOB_STRIP_LDS,              /* 0: border rows / columns of periodic plans on the round-4 kernels; 1 (default): ewa_strip_kernel by rule; 2: rows and columns on it always */
    JINC_KNOB_EDGE_COLS,              /* 0: border columns on the border kernels even where ewa_periodic_quad2_kernel's edge tiles could compute them (round 5) */
    JINC_KNOB_ROWPAIR_ROWS,           /* 0: border rows of filter sizes 11 .. 17 at 2x on ewa_direct_kernel's row strips instead of ewa_periodic_rowpair_kernel launches (round 5) */
    JINC_KNOB_COLPAIR,                /* border columns on ewa_colpair_kernel: 0 never, 1 (default) wherever configured, 3 filter sizes from 11 on only (round 5) */
    JINC_KNOB_UPLOAD_BOUNCE,          /* create-time table uploads: 1 (default) through the library's pinned buffer, 0 straight from the host vectors (round 6 A/B) */
    JINC_KNOB_COPY_THREADS,           /* CPU threads that copy a large pageable plane to / from the library's pinned buffers: 1 = the calling thread only; default 6 on hosts with 12 CPUs or more, 4 from 8 on, 2 from 4 on (round 6) */
    JINC_KNOB_STAGE_BANDS,            /* row bands a pageable plane is cut into between the CPU's copy and the DMA engine: default 4 (2 MiB each at least), 1 = whole planes (round 6) */
    JINC_KNOB_STAGE_DEFER_KB,         /* pageable frames of up to this many KiB of source in groups of 4 or more are copied to the library's pinned buffer when the group is launched (one job, one DMA copy per plane) instead of at submit: default 1536, 0 = never (round 6) */
    JINC_KNOB_COUNT
} jinc_knob;
JINC_API int jinc_debug_set_knob(int knob, double value);
JINC_API int jinc_debug_clear_knob(int knob);               /* knob < 0: every knob back to unset */
JINC_API int jinc_debug_get_knob(int knob, double *value);  /* 1: set (*value receives it), 0: unset, < 0: no such knob */
/* Chord patterns of the (fs - 1)-row x fs-column supports (chroma planes sited as MPEG-2 at 2x; csrc/kernels.h quad_span7 / kQuadSpan9*):
 * `spans` = per (kernel row ly, row phase q) the zero coefficients in front of / behind the row's span for both column phases, two bits each
 * (capped at 3) at 4 * (2 * ly + q) and 4 * (2 * ly + q) + 2.  Returns 1 / 2 if the kernels' compile-time pattern / its row-phase-swapped twin
 * leaves out no more than that, 0 if neither (all taps of the support then), -1 for a tap count without patterns.  Host only: no device call. */
JINC_API int jinc_debug_chord_pattern(int taps_per_row, uint64_t spans);
JINC_API const char *jinc_debug_knob_name(int knob);        /* lower-case name ("quad_rg"); NULL beyond the last knob */
/* Taps per axis the periodic interior kernels of `table` execute under the current kernel mode: the plan's filter size, or
 * the side of the trimmed support on integer planes (kernel mode 15 switches trimming off); 0 when the table has no
 * periodic interior or the instance has no device. */
JINC_API int jinc_filter_periodic_support(const jinc_filter *f, int table);
/* Taps per output sample the periodic interior kernels of `table` execute under the current kernel mode (bench.py: the VALU
 * fraction counts executed operations): side^2 of jinc_filter_periodic_support for the window / quad forms; for
 * ewa_periodic_rows_kernel (rows_kernel == 1) the sum of its per-kernel-row spans, averaged over the phases; rows_kernel == 2:
 * the direct kernel's interior (its trimmed support squared); 3: ewa_periodic_quad2_kernel (34 where its chord rows run on four taps);
 * 4: ewa_periodic_rowpair_kernel (the spans both phases p of a kernel row share). */
JINC_API double jinc_filter_periodic_taps(const jinc_filter *f, int table, int rows_kernel);
/* Name of the kernel that computes the interior of `table` under the current kernel mode (reports, profiles). */
JINC_API const char *jinc_filter_interior_kernel(const jinc_filter *f, int table);
/* Name of the kernel that computed the interior of `table` in the most recent frame call (the choice depends on the
 * batch size: the frame-lane kernel needs a batch). "" before the first call. */
JINC_API const char *jinc_filter_last_kernel(const jinc_filter *f, int table);
/* 1: the device's buffer range check covers the scalar offset, so ewa_direct_kernel and the strip border kernels are in
 * use; 0: it does not, and their plans run on the gather kernel (bench.py reports it: a silent fallback would show);
 * -1: host-only instance. */
JINC_API int jinc_filter_direct_premise(const jinc_filter *f);
/* Frames the look-ahead pipeline coalesces into one launch (what jinc_filter_set_pipeline[_group] settled on after its
 * automatic rule and the device-memory budget). */
JINC_API int jinc_filter_pipeline_group(const jinc_filter *f);
/* Shader clock WHILE other kernels run (bench.py: roofline.shader_clock_ghz): start launches eight single-lane samplers on
 * a stream of their own (one per XCD; they stamp the shader-clock and the 100 MHz real-time counters and sleep in between),
 * stop raises their flag, waits for them and reports min / median / max over the samplers of
 * d(shader ticks) / d(real-time ticks) x 0.1 GHz.  max_seconds (<= 120) bounds their life if stop is never called. */
typedef struct jinc_clock_sampler jinc_clock_sampler;
JINC_API int jinc_debug_clock_sampler_start(int device, double max_seconds, jinc_clock_sampler **out);
JINC_API int jinc_debug_clock_sampler_stop(jinc_clock_sampler *s, double *ghz_min, double *ghz_median, double *ghz_max);
/* What the hot kernels' instruction pair sustains on this part (bench.py: roofline.valu_pair_sustained_Tops): v_mul_f32 with
 * an SGPR coefficient + v_add_f32 onto one chain per lane, nothing else in the loop, the chip filled with waves_per_simd
 * waves per SIMD; Tops = multiplies + adds per second / 1e12, and the shader clock sampled beside the last launch. */
JINC_API int jinc_debug_valu_pair_probe(int device, int waves_per_simd, double *tops, double *shader_clock_ghz);
/* Interior kernel (of table 0) and frame count of the most recent kernel call of ANY filter instance in this process:
 * for tests that drive the plugin shell and cannot reach its jinc_filter handles. */
JINC_API const char *jinc_debug_last_call(int *nframes);
/* ... and that kernel's full instantiation (jinc_filter_last_instance of table 0 of the same call). */
JINC_API const char *jinc_debug_last_instance(void);
/* How the frames of the look-ahead pipeline left the device since the last reset, over ALL filter instances of this process:
 * written by the shader straight into pinned host planes (every destination plane of the group was pinned) or by DMA copies
 * (some plane was pageable); and how many host ranges the process-wide registry currently holds pinned.  For tests that
 * drive the plugin shell: several instances that share the host's frame pool must all keep the shader path. */
JINC_API int jinc_debug_transport_counts(long long *by_shader, long long *by_dma, long long *pinned_ranges, int reset);
/* ... and how many frames went through the library's OWN pinned buffers instead (register_host_buffers = 0: the caller's planes are
 * pageable and only the CPU touches them); reset together with the counts above. */
JINC_API long long jinc_debug_staged_frames(void);
/* host_copy.cpp's plane copy (rows of row_bytes bytes between two pitched buffers, cut into row ranges for the process-wide helper
 * threads when may_use_helpers and the plane is large): needs no device. */
/* CPUs the process may keep busy as host_copy.cpp counts them (affinity mask, cut down to the CFS quota of the process's cgroup):
 * what the helper pool of the plane copies is sized by. */
JINC_API int jinc_debug_usable_cpus(void);
JINC_API int jinc_debug_copy_rows(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t row_bytes, int rows,
                                  int may_use_helpers);
/* Border frame of exactly periodic plans: -1 (default) = by call size (strip kernels from ~5e9 taps per call on, one gather
 * launch below); 1 = rows and columns on the round-4 strip kernels (ewa_direct_kernel's row strips, ewa_colstrip_kernel or, in batches,
 * the frame-lane kernel), corners on the gather kernel; 2 = rows on the strip kernel, columns and corners on the gather kernel;
 * 3 = rows and columns on ewa_strip_kernel (round 5; filter sizes up to 9 at source step 1: the automatic choice there), corners on
 * the gather kernel; 4 = as 3, but the columns inside the interior kernel's edge tiles where that form exists (ewa_periodic_quad2_kernel
 * on integer planes: the automatic choice there); 0 = everything on the gather kernel (A/B measurements, tests). */
JINC_API int jinc_filter_set_border_strips(jinc_filter *f, int enable);
/* Which kernels computed the border frame of `table` in the most recent frame call, as bits: 1 gather kernel over the frame (or its
 * columns), 2 ewa_direct_kernel row strips, 4 ewa_colstrip_kernel, 8 frame-lane kernel over the columns, 16 / 32 ewa_strip_kernel
 * over the rows / the columns, 256 ewa_colpair_kernel over the columns, 128 the rows as launches of ewa_periodic_rowpair_kernel, 64 the columns inside the interior kernel's edge tiles (ewa_periodic_quad2_kernel, integer planes);
 * 0: no border launch recorded (plans whose border is not a strip frame). */
JINC_API int jinc_filter_last_border(const jinc_filter *f, int table);
/* 1: the border kernels run on a side stream concurrently with the interior kernel (fork/join by events around
 * every call); 0: all on the caller's stream, back to back; -1 (default): the side stream unless the call is so small
 * (below ~1e9 taps) that the fork/join costs more than it hides. */
JINC_API int jinc_filter_set_border_overlap(jinc_filter *f, int enable);

/* Test hook: runs the kernels' own sum -> sample conversion (clamp to [0, peak], round-half-even, store;
 * ref :581-584) on `n` caller-supplied fp32 sums on device `device` and returns the samples
 * (sample_bytes 1, 2 or 4).  Lets tests probe ties, bounds, NaN and infinities directly. */
JINC_API int jinc_debug_convert(const float *sums, void *out, int n, int sample_bytes, float peak, int device);

/* Test hook: 1 when the device's buffer range check covers the scalar offset of buffer loads (the premise of the
 * direct kernel's bounded segment fetches; probed once per device, the direct kernel is not used where it fails), 0 when
 * it does not, negative status when the probe could not run. */
JINC_API int jinc_debug_buffer_range_check(int device);
/* Process-wide A/B knob of the direct kernel's interior form (avisynth-jincresize_amd/csrc/kernel_direct_impl.inc,
 * DirectShape): 0 per-chain fetches, 2 row walk, 3 row walk with 8 columns per lane (where it applies), -1 automatic. */
JINC_API int jinc_debug_set_direct_shape(int shape);
/* DirectShape of the most recent interior launch of the direct kernel in this process (-1: none yet). */
JINC_API int jinc_debug_last_direct_shape(void);

/* ---- Kernel timing (benchmarks) ---------------------------------------------------------------
 * When enabled, every kernel launch made by jinc_filter_get_frame / jinc_filter_process_device is
 * bracketed by a pair of hipEvents recorded on the launch stream.  jinc_filter_kernel_times waits
 * for the recorded events, returns the accumulated device time (milliseconds) and launch count
 * of the periodic-interior kernel and of the gather kernel since the last call, and resets them. */
JINC_API int jinc_filter_set_profiling(jinc_filter *f, int enable);
JINC_API int jinc_filter_kernel_times(jinc_filter *f, double *periodic_ms, int *periodic_launches,
                                      double *gather_ms, int *gather_launches);

#ifdef __cplusplus
}
#endif
#endif /* JINCRESIZE_HIP_TEST_H */
