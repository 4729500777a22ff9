// tests/mock_avs/mock_host.cpp -- TEST INFRASTRUCTURE ONLY: a miniature AviSynth-like host that implements the API
// subset declared in tests/mock_avs/avisynth_c.h, plus a plain C interface (mock_*) through which
// tests/test_plugin_mock_host.py loads plugin/jincresize_avs.cpp, invokes its script functions with positional and
// named arguments, pulls frames and reads frame properties.  See the header for what this does and does not prove.
#include <sys/mman.h>
#include "avisynth_c.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

struct AVS_Map {
    std::map<std::string, int64_t> ints;
};

// The mock's private encoding of the clip format in AVS_VideoInfo::pixel_type (a real host has its own AVS_CS_* bits; the
// plugin never looks at pixel_type itself, only at what the host's avs_* accessors say).
static inline int pt_pack(int bits, int component_size, int num_components, int planar, int rgb, int sub_w, int sub_h) {
    return bits | component_size << 6 | num_components << 9 | (planar ? 1 : 0) << 12 | (rgb ? 1 : 0) << 13 | sub_w << 14 | sub_h << 16;
}
static inline int pt_bits(const AVS_VideoInfo* vi) { return vi->pixel_type & 63; }
static inline int pt_component_size(const AVS_VideoInfo* vi) { return (vi->pixel_type >> 6) & 7; }
static inline int pt_num_components(const AVS_VideoInfo* vi) { return (vi->pixel_type >> 9) & 7; }
static inline int pt_planar(const AVS_VideoInfo* vi) { return (vi->pixel_type >> 12) & 1; }
static inline int pt_rgb(const AVS_VideoInfo* vi) { return (vi->pixel_type >> 13) & 1; }
static inline int pt_sub_w(const AVS_VideoInfo* vi) { return (vi->pixel_type >> 14) & 3; }
static inline int pt_sub_h(const AVS_VideoInfo* vi) { return (vi->pixel_type >> 16) & 3; }

// One allocation per frame, planes at 64-byte aligned offsets (as AviSynth+ lays frames out).  Buffers of frames made by
// avs_new_video_frame_p come from -- and go back to -- the environment's frame pool (mock_env_set_frame_pool): every
// filter instance of a script draws from the same pool, so the same host buffer reaches different instances in turn.
// A frame buffer is an anonymous mapping of its own (whole pages that hold nothing else), as large frame allocations are in a real
// host: a plugin that pins frame buffers in place (JINCRESIZE_PIN_FRAMES) then maps no page that somebody else's data lives in.
struct FrameBuffer {
    unsigned char* base = nullptr;  // page-aligned
    size_t size = 0, mapped = 0;
    ~FrameBuffer() {
        if (base) munmap(base, mapped);
    }
};

struct AVS_VideoFrame {
    std::atomic<int> refs{1};
    AVS_ScriptEnvironment* counted_by = nullptr;  // frames made by avs_new_video_frame_p are counted while alive
    int nplanes = 0;
    int plane_id[4] = {0, 0, 0, 0};
    int pitch[4] = {0, 0, 0, 0}, row_size[4] = {0, 0, 0, 0}, height[4] = {0, 0, 0, 0};
    size_t offset[4] = {0, 0, 0, 0};
    FrameBuffer* buffer = nullptr;
    AVS_ScriptEnvironment* pool_of = nullptr;  // where the buffer goes when the frame dies (nullptr: deleted)
    AVS_Map props;
    unsigned char* plane(int i) const { return buffer->base + offset[i]; }
    ~AVS_VideoFrame();
};

struct Function {
    std::string name, params;
    AVS_ApplyFunc apply;
    void* user_data;
};

struct AVS_ScriptEnvironment {
    int interface_version = 10, interface_bugfix = 0;
    int cpu_flags = AVS_CPUF_SSE4_1 | AVS_CPUF_AVX2;
    std::vector<Function> functions;
    std::vector<std::unique_ptr<std::vector<AVS_Value>>> arg_arrays;  // storage of positional arrays built by avs_invoke
    std::atomic<long> live_frames{0}, live_clips{0};
    // frame pool: up to pool_limit idle buffers are kept for reuse (0: none, every frame gets a fresh allocation)
    std::mutex pool_mutex;
    std::vector<FrameBuffer*> pool;
    size_t pool_limit = 0;
    std::atomic<long> pool_reuses{0};
    ~AVS_ScriptEnvironment() {
        for (FrameBuffer* b : pool) delete b;
    }
};

AVS_VideoFrame::~AVS_VideoFrame() {
    if (!buffer) return;
    if (pool_of) {
        std::lock_guard<std::mutex> lock(pool_of->pool_mutex);
        if (pool_of->pool.size() < pool_of->pool_limit) {
            pool_of->pool.push_back(buffer);
            return;
        }
    }
    delete buffer;
}

struct AVS_Clip {
    std::atomic<int> refs{1};
    AVS_ScriptEnvironment* env = nullptr;
    AVS_VideoInfo vi{};
    // source clip
    std::vector<std::unique_ptr<AVS_VideoFrame>> frames;
    std::atomic<int> get_frame_calls{0};
    std::vector<std::atomic<int>> calls_of_frame;  // how often each frame was asked for
    // filter clip
    std::unique_ptr<AVS_FilterInfo> fi;
    std::mutex serialized;  // a filter that answers MT_SERIALIZED (3) is called by one thread at a time
};

namespace {

int frame_plane(const AVS_VideoFrame* f, int plane) {
    for (int i = 0; i < f->nplanes; ++i)
        if (f->plane_id[i] == plane) return i;
    return -1;
}

AVS_VideoFrame* make_frame(const AVS_VideoInfo* vi, int pitch_align, AVS_ScriptEnvironment* pool_env = nullptr) {
    static const int yuv[4] = {AVS_PLANAR_Y, AVS_PLANAR_U, AVS_PLANAR_V, AVS_PLANAR_A};
    static const int rgb[4] = {AVS_PLANAR_G, AVS_PLANAR_B, AVS_PLANAR_R, AVS_PLANAR_A};
    auto* f = new AVS_VideoFrame;
    f->nplanes = pt_num_components(vi);
    size_t total = 0;
    for (int i = 0; i < f->nplanes; ++i) {
        const bool chroma = !pt_rgb(vi) && (i == 1 || i == 2);
        const int w = chroma ? vi->width >> pt_sub_w(vi) : vi->width;
        const int h = chroma ? vi->height >> pt_sub_h(vi) : vi->height;
        f->plane_id[i] = (pt_rgb(vi) ? rgb : yuv)[i];
        f->row_size[i] = w * pt_component_size(vi);
        f->pitch[i] = (f->row_size[i] + pitch_align - 1) / pitch_align * pitch_align;
        f->height[i] = h;
        f->offset[i] = total;
        total += (static_cast<size_t>(f->pitch[i]) * h + 64 + 63) / 64 * 64;
    }
    if (pool_env) {
        std::lock_guard<std::mutex> lock(pool_env->pool_mutex);
        for (size_t k = 0; k < pool_env->pool.size(); ++k)
            if (pool_env->pool[k]->size == total) {
                f->buffer = pool_env->pool[k];
                pool_env->pool.erase(pool_env->pool.begin() + static_cast<std::ptrdiff_t>(k));
                ++pool_env->pool_reuses;
                break;
            }
        f->pool_of = pool_env;
    }
    if (!f->buffer) {
        f->buffer = new FrameBuffer;
        f->buffer->mapped = (total + 4095) / 4096 * 4096;
        void* m = mmap(nullptr, f->buffer->mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) {
            delete f->buffer;
            f->buffer = nullptr;
            delete f;
            return nullptr;
        }
        f->buffer->base = static_cast<unsigned char*>(m);
        f->buffer->size = total;
    }
    std::memset(f->buffer->base, 0xCD, total);
    return f;
}

}  // namespace

extern "C" {

AVS_Value avs_new_value_clip(AVS_Clip* clip) {
    AVS_Value v;
    v.type = 'c';
    v.array_size = 0;
    v.d.clip = clip;
    ++clip->refs;
    return v;
}

int avs_is_planar(const AVS_VideoInfo* vi) { return pt_planar(vi); }
int avs_is_rgb(const AVS_VideoInfo* vi) { return pt_rgb(vi); }
int avs_bits_per_component(const AVS_VideoInfo* vi) { return pt_bits(vi); }
int avs_component_size(const AVS_VideoInfo* vi) { return pt_component_size(vi); }
int avs_num_components(const AVS_VideoInfo* vi) { return pt_num_components(vi); }
int avs_get_plane_width_subsampling(const AVS_VideoInfo* vi, int plane) {
    return (plane == AVS_PLANAR_U || plane == AVS_PLANAR_V) ? pt_sub_w(vi) : 0;
}
int avs_get_plane_height_subsampling(const AVS_VideoInfo* vi, int plane) {
    return (plane == AVS_PLANAR_U || plane == AVS_PLANAR_V) ? pt_sub_h(vi) : 0;
}

int avs_check_version(AVS_ScriptEnvironment* env, int version) { return env->interface_version >= version ? 0 : -1; }
int64_t avs_get_env_property(AVS_ScriptEnvironment* env, int prop) {
    if (prop == AVS_AEP_INTERFACE_BUGFIX) return env->interface_bugfix;
    if (prop == AVS_AEP_INTERFACE_VERSION) return env->interface_version;
    return 0;
}
int avs_get_cpu_flags(AVS_ScriptEnvironment* env) { return env->cpu_flags; }

int avs_add_function(AVS_ScriptEnvironment* env, const char* name, const char* params, AVS_ApplyFunc apply, void* user_data) {
    env->functions.push_back({name, params, apply, user_data});
    return 0;
}

// Resolves positional + named arguments against the function's parameter string ("cii[src_left]f...") into the
// positional array the apply function indexes, with void values for what was not given; type letters are checked.
AVS_Value avs_invoke(AVS_ScriptEnvironment* env, const char* name, AVS_Value args, const char** arg_names) {
    const Function* fn = nullptr;
    for (const Function& f : env->functions)
        if (f.name == name) fn = &f;
    if (!fn) return avs_new_value_error("mock host: no such function");
    struct Param {
        std::string name;
        char type;
    };
    std::vector<Param> params;
    for (const char* p = fn->params.c_str(); *p;) {
        Param q;
        if (*p == '[') {
            const char* e = std::strchr(p, ']');
            q.name.assign(p + 1, e);
            p = e + 1;
        }
        q.type = *p++;
        params.push_back(q);
    }
    auto arr = std::make_unique<std::vector<AVS_Value>>(params.size());
    for (AVS_Value& v : *arr) v.type = 'v', v.array_size = 0, v.d.clip = nullptr;
    const int n = args.type == 'a' ? args.array_size : 1;
    size_t pos = 0;
    for (int i = 0; i < n; ++i) {
        const AVS_Value v = avs_array_elt(args, i);
        size_t slot = params.size();
        if (arg_names && arg_names[i]) {
            for (size_t k = 0; k < params.size(); ++k)
                if (params[k].name == arg_names[i]) slot = k;
            if (slot == params.size()) return avs_new_value_error("mock host: function does not have a named argument of that name");
        } else {
            slot = pos++;
            if (slot >= params.size()) return avs_new_value_error("mock host: too many arguments");
        }
        const char t = params[slot].type;
        const bool ok = (t == 'c' && v.type == 'c') || (t == 'i' && v.type == 'i') || (t == 'f' && (v.type == 'f' || v.type == 'i')) ||
                        (t == 's' && v.type == 's') || (t == 'b' && v.type == 'b');
        if (!ok) return avs_new_value_error("mock host: invalid arguments to function");
        (*arr)[slot] = v;
    }
    for (size_t k = 0; k < params.size(); ++k)
        if (params[k].name.empty() && (*arr)[k].type == 'v') return avs_new_value_error("mock host: missing positional argument");
    AVS_Value positional = avs_new_value_array(arr->data(), static_cast<int>(arr->size()));
    env->arg_arrays.push_back(std::move(arr));
    return fn->apply(env, positional, fn->user_data);
}

AVS_Clip* avs_new_c_filter(AVS_ScriptEnvironment* env, AVS_FilterInfo** fi, AVS_Value child, int store_child) {
    auto* clip = new AVS_Clip;
    clip->env = env;
    clip->fi = std::make_unique<AVS_FilterInfo>();
    std::memset(clip->fi.get(), 0, sizeof(AVS_FilterInfo));
    AVS_Clip* c = static_cast<AVS_Clip*>(child.d.clip);
    clip->fi->child = c;
    if (store_child) ++c->refs;
    clip->fi->vi = c->vi;
    clip->fi->env = env;
    clip->vi = c->vi;
    *fi = clip->fi.get();
    ++env->live_clips;
    return clip;
}

void avs_release_clip(AVS_Clip* clip) {
    if (!clip || --clip->refs > 0) return;
    if (clip->fi) {
        if (clip->fi->free_filter) clip->fi->free_filter(clip->fi.get());
        avs_release_clip(clip->fi->child);
    }
    --clip->env->live_clips;
    delete clip;
}

AVS_VideoFrame* avs_get_frame(AVS_Clip* clip, int n) {
    if (clip->fi) {
        // a C filter without a get_frame callback passes the request through to its child
        if (!clip->fi->get_frame) return avs_get_frame(clip->fi->child, n);
        // MT_SERIALIZED (3): the host's worker threads take turns at the one instance.  (MT_MULTI_INSTANCE (2): a real host
        // creates an instance per thread -- here the test invokes the function once per thread.)
        // As AviSynth+'s C_VideoFilter::GetFrame: when the filter set fi->error the host throws BEFORE it takes ownership of the
        // returned frame -- a frame returned beside an error is never released (and shows up in mock_live_frames).
        auto call = [&]() -> AVS_VideoFrame* {
            clip->fi->error = nullptr;
            AVS_VideoFrame* f = clip->fi->get_frame(clip->fi.get(), n);
            return clip->fi->error ? nullptr : f;
        };
        if (clip->fi->set_cache_hints && clip->fi->set_cache_hints(clip->fi.get(), AVS_CACHE_GET_MTMODE, 0) == 3) {
            std::lock_guard<std::mutex> lock(clip->serialized);
            return call();
        }
        return call();
    }
    ++clip->get_frame_calls;
    if (n < 0 || n >= static_cast<int>(clip->frames.size())) return nullptr;
    ++clip->calls_of_frame[static_cast<size_t>(n)];
    ++clip->frames[n]->refs;  // the source keeps its own reference
    return clip->frames[n].get();
}

AVS_VideoFrame* avs_new_video_frame_p(AVS_ScriptEnvironment* env, const AVS_VideoInfo* vi, const AVS_VideoFrame* prop_src) {
    AVS_VideoFrame* f = make_frame(vi, 64, env);
    if (prop_src) f->props = prop_src->props;
    f->counted_by = env;
    ++env->live_frames;
    return f;
}

void avs_release_video_frame(AVS_VideoFrame* frame) {
    if (!frame) return;
    if (--frame->refs > 0) return;  // source-owned frames never reach 0 here (the clip holds one reference)
    if (frame->counted_by) --frame->counted_by->live_frames;
    delete frame;
}

int avs_get_pitch_p(const AVS_VideoFrame* f, int plane) { const int i = frame_plane(f, plane); return i < 0 ? 0 : f->pitch[i]; }
int avs_get_row_size_p(const AVS_VideoFrame* f, int plane) { const int i = frame_plane(f, plane); return i < 0 ? 0 : f->row_size[i]; }
int avs_get_height_p(const AVS_VideoFrame* f, int plane) { const int i = frame_plane(f, plane); return i < 0 ? 0 : f->height[i]; }
const unsigned char* avs_get_read_ptr_p(const AVS_VideoFrame* f, int plane) {
    const int i = frame_plane(f, plane);
    return i < 0 ? nullptr : f->plane(i);
}
unsigned char* avs_get_write_ptr_p(const AVS_VideoFrame* f, int plane) {
    const int i = frame_plane(f, plane);
    return i < 0 ? nullptr : f->plane(i);
}

const AVS_Map* avs_get_frame_props_ro(AVS_ScriptEnvironment*, const AVS_VideoFrame* frame) { return &frame->props; }
AVS_Map* avs_get_frame_props_rw(AVS_ScriptEnvironment*, AVS_VideoFrame* frame) { return &frame->props; }
char avs_prop_get_type(AVS_ScriptEnvironment*, const AVS_Map* map, const char* key) { return map->ints.count(key) ? 'i' : 'u'; }
int64_t avs_prop_get_int(AVS_ScriptEnvironment*, const AVS_Map* map, const char* key, int, int* error) {
    auto it = map->ints.find(key);
    if (error) *error = it == map->ints.end();
    return it == map->ints.end() ? 0 : it->second;
}
int avs_prop_set_int(AVS_ScriptEnvironment*, AVS_Map* map, const char* key, int64_t value, int) {
    map->ints[key] = value;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// C interface for the Python tests
// ---------------------------------------------------------------------------------------------------------------
const char* avisynth_c_plugin_init(AVS_ScriptEnvironment* env);  // the plugin under test, linked into this library

#define MOCK_API __attribute__((visibility("default")))

MOCK_API AVS_ScriptEnvironment* mock_env_new(int interface_version, int interface_bugfix, int cpu_flags) {
    auto* env = new AVS_ScriptEnvironment;
    env->interface_version = interface_version;
    env->interface_bugfix = interface_bugfix;
    if (cpu_flags >= 0) env->cpu_flags = cpu_flags;
    return env;
}
MOCK_API void mock_env_free(AVS_ScriptEnvironment* env) { delete env; }
MOCK_API const char* mock_load_plugin(AVS_ScriptEnvironment* env) { return avisynth_c_plugin_init(env); }
MOCK_API int mock_function_count(AVS_ScriptEnvironment* env) { return static_cast<int>(env->functions.size()); }
MOCK_API const char* mock_function_name(AVS_ScriptEnvironment* env, int i) { return env->functions[i].name.c_str(); }
MOCK_API const char* mock_function_params(AVS_ScriptEnvironment* env, int i) { return env->functions[i].params.c_str(); }
MOCK_API long mock_live_frames(AVS_ScriptEnvironment* env) { return env->live_frames; }
MOCK_API long mock_live_clips(AVS_ScriptEnvironment* env) { return env->live_clips; }

// A source clip of `num_frames` frames whose planes the test fills through mock_source_plane.
// chroma_location >= 0 attaches the int property _ChromaLocation to every frame.
MOCK_API AVS_Clip* mock_source_new(AVS_ScriptEnvironment* env, int width, int height, int bits, int component_size, int num_components,
                                   int planar, int rgb, int sub_w, int sub_h, int num_frames, int chroma_location, int pitch_align) {
    auto* clip = new AVS_Clip;
    clip->env = env;
    AVS_VideoInfo& vi = clip->vi;
    std::memset(&vi, 0, sizeof vi);
    vi.width = width, vi.height = height, vi.fps_numerator = 24, vi.fps_denominator = 1, vi.num_frames = num_frames;
    vi.pixel_type = pt_pack(bits, component_size, num_components, planar, rgb, sub_w, sub_h);
    for (int n = 0; n < num_frames; ++n) {
        clip->frames.emplace_back(make_frame(&vi, pitch_align > 0 ? pitch_align : 64));
        if (chroma_location >= 0) clip->frames.back()->props.ints["_ChromaLocation"] = chroma_location;
    }
    clip->calls_of_frame = std::vector<std::atomic<int>>(static_cast<size_t>(num_frames));
    ++env->live_clips;
    return clip;
}
MOCK_API unsigned char* mock_frame_plane(AVS_VideoFrame* f, int index, int* pitch, int* row_size, int* height) {
    if (index < 0 || index >= f->nplanes) return nullptr;
    *pitch = f->pitch[index], *row_size = f->row_size[index], *height = f->height[index];
    return f->plane(index);
}
MOCK_API AVS_VideoFrame* mock_source_frame(AVS_Clip* clip, int n) { return clip->frames[n].get(); }
MOCK_API int mock_source_get_frame_calls(AVS_Clip* clip) { return clip->get_frame_calls; }
// How often frame n of a source clip was asked for (a look-ahead that fetches a child frame twice shows here).
MOCK_API int mock_source_calls_of_frame(AVS_Clip* clip, int n) {
    return n >= 0 && n < static_cast<int>(clip->calls_of_frame.size()) ? clip->calls_of_frame[static_cast<size_t>(n)].load() : -1;
}
// Frame pool of the environment: up to `buffers` idle frame buffers are kept and handed out again by avs_new_video_frame_p.
MOCK_API void mock_env_set_frame_pool(AVS_ScriptEnvironment* env, int buffers) { env->pool_limit = buffers > 0 ? static_cast<size_t>(buffers) : 0; }
MOCK_API long mock_env_pool_reuses(AVS_ScriptEnvironment* env) { return env->pool_reuses; }

// Invokes a script function: positional (clip, width, height) + named arguments.
// kinds[i]: 'i' -> ivals[i], 'f' -> fvals[i], 's' -> svals[i].  Returns a heap AVS_Value (mock_value_free).
MOCK_API AVS_Value* mock_invoke(AVS_ScriptEnvironment* env, const char* function, AVS_Clip* clip, int width, int height, int nnamed,
                                const char** names, const char* kinds, const int* ivals, const double* fvals, const char** svals) {
    std::vector<AVS_Value> vals;
    std::vector<const char*> nm;
    AVS_Value c;
    c.type = 'c', c.array_size = 0, c.d.clip = clip;
    vals.push_back(c), nm.push_back(nullptr);
    vals.push_back(avs_new_value_int(width)), nm.push_back(nullptr);
    vals.push_back(avs_new_value_int(height)), nm.push_back(nullptr);
    for (int i = 0; i < nnamed; ++i) {
        if (kinds[i] == 'i') vals.push_back(avs_new_value_int(ivals[i]));
        else if (kinds[i] == 'f') vals.push_back(avs_new_value_float(static_cast<float>(fvals[i])));
        else vals.push_back(avs_new_value_string(svals[i]));
        nm.push_back(names[i]);
    }
    auto* out = new AVS_Value(avs_invoke(env, function, avs_new_value_array(vals.data(), static_cast<int>(vals.size())), nm.data()));
    return out;
}
MOCK_API const char* mock_value_error(const AVS_Value* v) { return v->type == 'e' ? v->d.string : nullptr; }
MOCK_API AVS_Clip* mock_value_clip(const AVS_Value* v) { return v->type == 'c' ? static_cast<AVS_Clip*>(v->d.clip) : nullptr; }
MOCK_API void mock_value_free(AVS_Value* v) { delete v; }

MOCK_API void mock_clip_info(AVS_Clip* clip, int* width, int* height, int* num_frames) {
    const AVS_VideoInfo& vi = clip->fi ? clip->fi->vi : clip->vi;
    *width = vi.width, *height = vi.height, *num_frames = vi.num_frames;
}
MOCK_API AVS_VideoFrame* mock_clip_get_frame(AVS_Clip* clip, int n) { return avs_get_frame(clip, n); }
MOCK_API const char* mock_clip_error(AVS_Clip* clip) { return clip->fi ? clip->fi->error : nullptr; }
MOCK_API int mock_clip_mt_mode(AVS_Clip* clip) {
    return clip->fi && clip->fi->set_cache_hints ? clip->fi->set_cache_hints(clip->fi.get(), AVS_CACHE_GET_MTMODE, 0) : -1;
}
MOCK_API void mock_clip_release(AVS_Clip* clip) { avs_release_clip(clip); }
MOCK_API void mock_source_release(AVS_Clip* clip) {
    if (--clip->refs > 0) return;
    --clip->env->live_clips;
    delete clip;
}
MOCK_API int mock_frame_prop_int(AVS_VideoFrame* f, const char* key, long long* value) {
    auto it = f->props.ints.find(key);
    if (it == f->props.ints.end()) return 0;
    *value = it->second;
    return 1;
}
MOCK_API void mock_frame_release(AVS_VideoFrame* f) { avs_release_video_frame(f); }

}  // extern "C"
