/* plugin/compat/avisynth_c.h -- COMPATIBILITY DECLARATION of the part of the AviSynth+ C API that
 * plugin/jincresize_avs.cpp uses (the symbol list of SURVEY.md 8(b)).
 *
 * The AviSynth+ SDK header is not part of this repository or its build image.  This file is written from the API's
 * public names and call shapes so that the plugin can be compiled here -- into plugin/lib/libjincresize.so by
 * __graft_entry__.build(), and together with the mock host of tests/mock_avs/ for the plugin tests.  It is never used
 * to compile any file of /root/reference.  When building against a real AviSynth+ installation, put the SDK's include
 * directory FIRST on the include path so that its avisynth_c.h replaces this one; nothing in the plugin depends on this
 * file's details.
 *
 * Struct layouts follow the upstream field order as far as the author knows it (AVS_VideoInfo, AVS_Value,
 * AVS_FilterInfo: the plugin touches vi.width / vi.height / vi.num_frames, fi->child / vi / env / get_frame /
 * set_cache_hints / free_filter / error / user_data, and AVS_Value only through the avs_* accessors).
 * TO BE VERIFIED AGAINST UPSTREAM avisynth_c.h before trusting a binary built with this file:
 *   - numeric values of AVS_PLANAR_*, AVS_CPUF_SSE4_1 / AVX2 / AVX512F, AVS_CACHE_GET_MTMODE, AVS_AEP_INTERFACE_BUGFIX
 *   - field order and types of AVS_VideoInfo (esp. the audio fields before image_type), AVS_Value (type letters, the
 *     64-bit members of newer interface versions), AVS_FilterInfo
 *   - which avs_* accessors are inline in the SDK header and which are imported from the host library (here all that
 *     need host state are plain external functions; the value accessors are inline)
 *   - AVSC_CC (stdcall on 32-bit Windows) and the AVSC_EXPORT / avisynth_c_plugin_init declaration
 */
#ifndef JINCRESIZE_COMPAT_AVISYNTH_C_H
#define JINCRESIZE_COMPAT_AVISYNTH_C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVSC_CC
#define AVSC_EXPORT __attribute__((visibility("default")))
#define AVSC_API(ret, name) ret AVSC_CC name

enum { AVS_PLANAR_Y = 1 << 0, AVS_PLANAR_U = 1 << 1, AVS_PLANAR_V = 1 << 2, AVS_PLANAR_ALIGNED = 1 << 3,
       AVS_PLANAR_A = 1 << 4, AVS_PLANAR_R = 1 << 5, AVS_PLANAR_G = 1 << 6, AVS_PLANAR_B = 1 << 7 };
enum { AVS_CPUF_SSE4_1 = 0x400, AVS_CPUF_AVX2 = 0x2000, AVS_CPUF_AVX512F = 0x10000 };
enum { AVS_CACHE_GET_MTMODE = 509 };
enum { AVS_AEP_INTERFACE_VERSION = 50, AVS_AEP_INTERFACE_BUGFIX = 51 };

typedef struct AVS_Clip AVS_Clip;
typedef struct AVS_ScriptEnvironment AVS_ScriptEnvironment;
typedef struct AVS_VideoFrame AVS_VideoFrame;
typedef struct AVS_Map AVS_Map;

typedef struct AVS_VideoInfo {
    int width, height; /* width = 0 means no video */
    unsigned fps_numerator, fps_denominator;
    int num_frames;
    int pixel_type; /* opaque here: only the host's avs_is_* / avs_bits_per_component / ... interpret it */
    int audio_samples_per_second; /* 0 means no audio */
    int sample_type;
    int64_t num_audio_samples;
    int nchannels;
    int image_type;
} AVS_VideoInfo;

typedef struct AVS_Value {
    short type; /* 'a'rray, 'c'lip, 'b'ool, 'i'nt, 'f'loat, 's'tring, 'v'oid, 'e'rror ('l'ong, 'd'ouble, fu'n'ction in newer hosts) */
    short array_size;
    union {
        void* clip; /* do not use directly */
        char boolean;
        int integer;
        float floating_pt;
        const char* string;
        const struct AVS_Value* array;
        void* function;
        int64_t longlong;
        double double_pt;
    } d;
} AVS_Value;

typedef struct AVS_FilterInfo AVS_FilterInfo;
struct AVS_FilterInfo {
    AVS_Clip* child;
    AVS_VideoInfo vi;
    AVS_ScriptEnvironment* env;
    AVS_VideoFrame*(AVSC_CC* get_frame)(AVS_FilterInfo*, int n);
    int(AVSC_CC* get_parity)(AVS_FilterInfo*, int n);
    int(AVSC_CC* get_audio)(AVS_FilterInfo*, void* buf, int64_t start, int64_t count);
    int(AVSC_CC* set_cache_hints)(AVS_FilterInfo*, int cachehints, int frame_range);
    void(AVSC_CC* free_filter)(AVS_FilterInfo*);
    const char* error;
    void* user_data;
};

typedef AVS_Value(AVSC_CC* AVS_ApplyFunc)(AVS_ScriptEnvironment*, AVS_Value args, void* user_data);

/* values */
static inline int avs_defined(AVS_Value v) { return v.type != 'v'; }
static inline int avs_is_clip(AVS_Value v) { return v.type == 'c'; }
static inline int avs_is_error(AVS_Value v) { return v.type == 'e'; }
static inline int avs_as_int(AVS_Value v) { return v.d.integer; }
static inline double avs_as_float(AVS_Value v) { return v.type == 'i' ? v.d.integer : v.d.floating_pt; }
static inline const char* avs_as_string(AVS_Value v) { return v.type == 's' || v.type == 'e' ? v.d.string : 0; }
static inline const char* avs_as_error(AVS_Value v) { return v.type == 'e' ? v.d.string : 0; }
static inline AVS_Value avs_array_elt(AVS_Value v, int index) { return v.type == 'a' ? v.d.array[index] : v; }
static inline AVS_Value avs_new_value_int(int v0) { AVS_Value v; v.type = 'i'; v.array_size = 0; v.d.integer = v0; return v; }
static inline AVS_Value avs_new_value_float(float v0) { AVS_Value v; v.type = 'f'; v.array_size = 0; v.d.floating_pt = v0; return v; }
static inline AVS_Value avs_new_value_string(const char* v0) { AVS_Value v; v.type = 's'; v.array_size = 0; v.d.string = v0; return v; }
static inline AVS_Value avs_new_value_error(const char* v0) { AVS_Value v; v.type = 'e'; v.array_size = 0; v.d.string = v0; return v; }
static inline AVS_Value avs_new_value_array(AVS_Value* v0, int size) { AVS_Value v; v.type = 'a'; v.array_size = (short)size; v.d.array = v0; return v; }
AVS_Value avs_new_value_clip(AVS_Clip* clip); /* takes a reference */

/* video info */
int avs_is_planar(const AVS_VideoInfo* vi);
int avs_is_rgb(const AVS_VideoInfo* vi);
int avs_bits_per_component(const AVS_VideoInfo* vi);
int avs_component_size(const AVS_VideoInfo* vi);
int avs_num_components(const AVS_VideoInfo* vi);
int avs_get_plane_width_subsampling(const AVS_VideoInfo* vi, int plane);
int avs_get_plane_height_subsampling(const AVS_VideoInfo* vi, int plane);

/* environment */
int avs_check_version(AVS_ScriptEnvironment* env, int version); /* 0 = the host offers at least `version` */
int64_t avs_get_env_property(AVS_ScriptEnvironment* env, int prop);
int avs_get_cpu_flags(AVS_ScriptEnvironment* env);
int avs_add_function(AVS_ScriptEnvironment* env, const char* name, const char* params, AVS_ApplyFunc apply, void* user_data);
AVS_Value avs_invoke(AVS_ScriptEnvironment* env, const char* name, AVS_Value args, const char** arg_names);

/* clips and frames */
AVS_Clip* avs_new_c_filter(AVS_ScriptEnvironment* env, AVS_FilterInfo** fi, AVS_Value child, int store_child);
void avs_release_clip(AVS_Clip* clip);
AVS_VideoFrame* avs_get_frame(AVS_Clip* clip, int n);
AVS_VideoFrame* avs_new_video_frame_p(AVS_ScriptEnvironment* env, const AVS_VideoInfo* vi, const AVS_VideoFrame* prop_src);
void avs_release_video_frame(AVS_VideoFrame* frame);
int avs_get_pitch_p(const AVS_VideoFrame* frame, int plane);
int avs_get_row_size_p(const AVS_VideoFrame* frame, int plane);
int avs_get_height_p(const AVS_VideoFrame* frame, int plane);
const unsigned char* avs_get_read_ptr_p(const AVS_VideoFrame* frame, int plane);
unsigned char* avs_get_write_ptr_p(const AVS_VideoFrame* frame, int plane);

/* frame properties */
const AVS_Map* avs_get_frame_props_ro(AVS_ScriptEnvironment* env, const AVS_VideoFrame* frame);
AVS_Map* avs_get_frame_props_rw(AVS_ScriptEnvironment* env, AVS_VideoFrame* frame);
char avs_prop_get_type(AVS_ScriptEnvironment* env, const AVS_Map* map, const char* key); /* 'i', 'f', 's', 'u'nset ... */
int64_t avs_prop_get_int(AVS_ScriptEnvironment* env, const AVS_Map* map, const char* key, int index, int* error);
int avs_prop_set_int(AVS_ScriptEnvironment* env, AVS_Map* map, const char* key, int64_t value, int append);

#ifdef __cplusplus
}
#endif
#endif
