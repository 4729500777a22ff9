// mock_vs_host.cpp -- a miniature VapourSynth (API 4) host for tests/test_plugin_vs_mock_host.py: implements the subset of
// VSAPI / VSPLUGINAPI that plugin/jincresize_vs.cpp uses (plugin/compat/VapourSynth4.h) -- maps with typed entries and
// errors, reference-counted nodes and frames with planes, strides and properties, a function registry that checks
// arguments against the registered signature strings, source nodes, and the two-step frame request protocol (arInitial ->
// requestFrameFilter -> arAllFramesReady -> getFrameFilter).  Test infrastructure: it proves the shell's logic, not binary
// compatibility with a real VapourSynth core.
#include "VapourSynth4.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

struct MapValue {
    int type = ptUnset;
    std::vector<int64_t> ints;
    std::vector<double> floats;
    std::vector<std::string> data;
    std::vector<VSNode*> nodes;
};

struct VSMap {
    std::map<std::string, MapValue> kv;
    bool has_error = false;
    std::string error;
};

struct VSFrame {
    VSVideoFormat format{};
    int width = 0, height = 0;
    std::vector<uint8_t> buf[3];
    ptrdiff_t stride[3] = {0, 0, 0};
    int pw[3] = {0, 0, 0}, ph[3] = {0, 0, 0};
    VSMap props;
    long refs = 1;
};

struct VSCore {
    long live_frames = 0, live_nodes = 0;
    int stride_align = 64;
    // Plane buffers of dead frames, kept for reuse by size for as long as the core lives -- as VapourSynth's own memory pool
    // recycles frame buffers.  A filter that pins frame memory in place (JINCRESIZE_PIN_FRAMES) relies on exactly that: the
    // host must not free pinned buffers while the filter instance lives.
    std::multimap<size_t, std::vector<uint8_t>> pool;
    long pool_reuses = 0;
};

struct VSFrameContext {
    std::map<std::pair<VSNode*, int>, const VSFrame*> ready;  // frames requested in arInitial, available in arAllFramesReady
    std::string error;
    bool has_error = false;
};

struct VSNode {
    VSCore* core = nullptr;
    VSVideoInfo vi{};
    long refs = 1;
    // source node: owns its frames
    std::vector<VSFrame*> frames;
    int get_frame_calls = 0;
    // filter node
    VSFilterGetFrame get_frame = nullptr;
    VSFilterFree free_fn = nullptr;
    void* instance = nullptr;
    int filter_mode = 0;
    std::vector<VSNode*> deps;
    std::string name;
};

struct VSPlugin {
    std::string identifier, ns, name;
    int api_version = 0;
    struct Fn {
        std::string name, args, ret;
        VSPublicFunction fn;
        void* data;
    };
    std::vector<Fn> fns;
};

namespace {
VSCore* g_core_of_api = nullptr;  // the VSAPI entry points carry no core for frames / nodes: one mock core at a time

VSFrame* new_frame(VSCore* core, const VSVideoFormat& f, int w, int h) {
    VSFrame* fr = new VSFrame;
    fr->format = f;
    fr->width = w;
    fr->height = h;
    for (int p = 0; p < f.numPlanes; ++p) {
        fr->pw[p] = p ? w >> f.subSamplingW : w;
        fr->ph[p] = p ? h >> f.subSamplingH : h;
        const int a = core->stride_align;
        fr->stride[p] = (static_cast<ptrdiff_t>(fr->pw[p]) * f.bytesPerSample + a - 1) / a * a;
        const size_t bytes = static_cast<size_t>(fr->stride[p]) * fr->ph[p] + 64;
        auto it = core->pool.find(bytes);
        if (it != core->pool.end()) {
            fr->buf[p] = std::move(it->second);
            core->pool.erase(it);
            ++core->pool_reuses;
            std::fill(fr->buf[p].begin(), fr->buf[p].end(), static_cast<uint8_t>(0xCD));
        } else {
            fr->buf[p].assign(bytes, 0xCD);
        }
    }
    ++core->live_frames;
    return fr;
}

void frame_unref(const VSFrame* cf) {
    VSFrame* f = const_cast<VSFrame*>(cf);
    if (f && --f->refs == 0) {
        --g_core_of_api->live_frames;
        for (int p = 0; p < 3; ++p)
            if (!f->buf[p].empty()) {
                const size_t bytes = f->buf[p].size();
                g_core_of_api->pool.emplace(bytes, std::move(f->buf[p]));
            }
        delete f;
    }
}

// ---- VSAPI entry points ----
void VS_CC api_createVideoFilter(VSMap* out, const char* name, const VSVideoInfo* vi, VSFilterGetFrame getFrame, VSFilterFree free_fn, int filterMode,
                                 const VSFilterDependency* deps, int numDeps, void* instanceData, VSCore* core) {
    VSNode* n = new VSNode;
    n->core = core;
    n->vi = *vi;
    n->get_frame = getFrame;
    n->free_fn = free_fn;
    n->instance = instanceData;
    n->filter_mode = filterMode;
    n->name = name;
    for (int i = 0; i < numDeps; ++i) n->deps.push_back(deps[i].source);
    ++core->live_nodes;
    MapValue v;
    v.type = ptVideoNode;
    v.nodes.push_back(n);  // the map owns this reference
    out->kv["clip"] = v;
}
void VS_CC api_freeNode(VSNode* node);
VSNode* VS_CC api_addNodeRef(VSNode* node) {
    ++node->refs;
    return node;
}
const VSVideoInfo* VS_CC api_getVideoInfo(VSNode* node) { return &node->vi; }
VSFrame* VS_CC api_newVideoFrame(const VSVideoFormat* format, int width, int height, const VSFrame* propSrc, VSCore* core) {
    VSFrame* f = new_frame(core, *format, width, height);
    if (propSrc) f->props.kv = propSrc->props.kv;
    return f;
}
void VS_CC api_freeFrame(const VSFrame* f) { frame_unref(f); }
const VSMap* VS_CC api_getFramePropertiesRO(const VSFrame* f) { return &f->props; }
VSMap* VS_CC api_getFramePropertiesRW(VSFrame* f) { return &f->props; }
ptrdiff_t VS_CC api_getStride(const VSFrame* f, int plane) { return f->stride[plane]; }
const uint8_t* VS_CC api_getReadPtr(const VSFrame* f, int plane) { return f->buf[plane].data(); }
uint8_t* VS_CC api_getWritePtr(VSFrame* f, int plane) { return f->buf[plane].data(); }
const VSVideoFormat* VS_CC api_getVideoFrameFormat(const VSFrame* f) { return &f->format; }
int VS_CC api_getFrameWidth(const VSFrame* f, int plane) { return f->pw[plane]; }
int VS_CC api_getFrameHeight(const VSFrame* f, int plane) { return f->ph[plane]; }

const VSFrame* produce(VSNode* node, int n, std::string& error, const VSAPI* api);

const VSFrame* VS_CC api_getFrame(int n, VSNode* node, char* errorMsg, int bufSize);
const VSFrame* VS_CC api_getFrameFilter(int n, VSNode* node, VSFrameContext* ctx) {
    auto it = ctx->ready.find({node, n});
    if (it == ctx->ready.end() || !it->second) return nullptr;
    VSFrame* f = const_cast<VSFrame*>(it->second);
    ++f->refs;  // the caller owns a reference
    return f;
}
const VSAPI* g_api = nullptr;
void VS_CC api_requestFrameFilter(int n, VSNode* node, VSFrameContext* ctx) {
    std::string err;
    const VSFrame* f = produce(node, n, err, g_api);  // the mock resolves requests at once
    if (!f) {
        ctx->has_error = true;
        ctx->error = err.empty() ? "frame not available" : err;
    }
    ctx->ready[{node, n}] = f;  // the context owns this reference
}
void VS_CC api_setFilterError(const char* msg, VSFrameContext* ctx) {
    ctx->has_error = true;
    ctx->error = msg ? msg : "";
}
VSMap* VS_CC api_createMap(void) { return new VSMap; }
void release_map_nodes(VSMap* m) {
    for (auto& kv : m->kv)
        for (VSNode* n : kv.second.nodes) api_freeNode(n);
}
void VS_CC api_freeMap(VSMap* m) {
    if (!m) return;
    release_map_nodes(m);
    delete m;
}
void VS_CC api_mapSetError(VSMap* m, const char* msg) {
    release_map_nodes(m);
    m->kv.clear();
    m->has_error = true;
    m->error = msg ? msg : "";
}
const char* VS_CC api_mapGetError(const VSMap* m) { return m->has_error ? m->error.c_str() : nullptr; }
int VS_CC api_mapNumElements(const VSMap* m, const char* key) {
    auto it = m->kv.find(key);
    if (it == m->kv.end()) return -1;
    const MapValue& v = it->second;
    return static_cast<int>(v.ints.size() + v.floats.size() + v.data.size() + v.nodes.size());
}
int VS_CC api_mapGetType(const VSMap* m, const char* key) {
    auto it = m->kv.find(key);
    return it == m->kv.end() ? ptUnset : it->second.type;
}
const MapValue* find(const VSMap* m, const char* key, int type, int index, size_t count_of, int* error) {
    auto it = m->kv.find(key);
    if (it == m->kv.end()) {
        if (error) *error = 1;  // peUnset
        return nullptr;
    }
    if (it->second.type != type) {
        if (error) *error = 2;  // peType
        return nullptr;
    }
    (void)count_of;
    (void)index;
    if (error) *error = 0;
    return &it->second;
}
int64_t VS_CC api_mapGetInt(const VSMap* m, const char* key, int index, int* error) {
    const MapValue* v = find(m, key, ptInt, index, 0, error);
    if (!v || index < 0 || index >= static_cast<int>(v->ints.size())) {
        if (v && error) *error = 4;  // peIndex
        return 0;
    }
    return v->ints[static_cast<size_t>(index)];
}
int VS_CC api_mapSetInt(VSMap* m, const char* key, int64_t i, int append) {
    MapValue& v = m->kv[key];
    if (append == maReplace || v.type != ptInt) v = MapValue{};
    v.type = ptInt;
    v.ints.push_back(i);
    return 0;
}
double VS_CC api_mapGetFloat(const VSMap* m, const char* key, int index, int* error) {
    const MapValue* v = find(m, key, ptFloat, index, 0, error);
    if (!v || index < 0 || index >= static_cast<int>(v->floats.size())) {
        if (v && error) *error = 4;
        return 0.0;
    }
    return v->floats[static_cast<size_t>(index)];
}
int VS_CC api_mapSetFloat(VSMap* m, const char* key, double d, int append) {
    MapValue& v = m->kv[key];
    if (append == maReplace || v.type != ptFloat) v = MapValue{};
    v.type = ptFloat;
    v.floats.push_back(d);
    return 0;
}
const char* VS_CC api_mapGetData(const VSMap* m, const char* key, int index, int* error) {
    const MapValue* v = find(m, key, ptData, index, 0, error);
    if (!v || index < 0 || index >= static_cast<int>(v->data.size())) {
        if (v && error) *error = 4;
        return nullptr;
    }
    return v->data[static_cast<size_t>(index)].c_str();
}
int VS_CC api_mapGetDataSize(const VSMap* m, const char* key, int index, int* error) {
    const MapValue* v = find(m, key, ptData, index, 0, error);
    if (!v || index < 0 || index >= static_cast<int>(v->data.size())) return -1;
    return static_cast<int>(v->data[static_cast<size_t>(index)].size());
}
int VS_CC api_mapSetData(VSMap* m, const char* key, const char* data, int size, int, int append) {
    MapValue& v = m->kv[key];
    if (append == maReplace || v.type != ptData) v = MapValue{};
    v.type = ptData;
    v.data.emplace_back(data, size < 0 ? std::strlen(data) : static_cast<size_t>(size));
    return 0;
}
VSNode* VS_CC api_mapGetNode(const VSMap* m, const char* key, int index, int* error) {
    const MapValue* v = find(m, key, ptVideoNode, index, 0, error);
    if (!v || index < 0 || index >= static_cast<int>(v->nodes.size())) {
        if (v && error) *error = 4;
        return nullptr;
    }
    return api_addNodeRef(v->nodes[static_cast<size_t>(index)]);  // the caller owns a reference
}
int VS_CC api_mapSetNode(VSMap* m, const char* key, VSNode* node, int append) {
    MapValue& v = m->kv[key];
    if (append == maReplace || v.type != ptVideoNode) v = MapValue{};
    v.type = ptVideoNode;
    v.nodes.push_back(api_addNodeRef(node));
    return 0;
}
int VS_CC api_mapConsumeNode(VSMap* m, const char* key, VSNode* node, int append) {
    const int r = api_mapSetNode(m, key, node, append);
    api_freeNode(node);
    return r;
}

VSAPI make_api() {
    VSAPI a;
    std::memset(&a, 0, sizeof a);
    a.createVideoFilter = api_createVideoFilter;
    a.freeNode = api_freeNode;
    a.addNodeRef = api_addNodeRef;
    a.getVideoInfo = api_getVideoInfo;
    a.newVideoFrame = api_newVideoFrame;
    a.freeFrame = api_freeFrame;
    a.getFramePropertiesRO = api_getFramePropertiesRO;
    a.getFramePropertiesRW = api_getFramePropertiesRW;
    a.getStride = api_getStride;
    a.getReadPtr = api_getReadPtr;
    a.getWritePtr = api_getWritePtr;
    a.getVideoFrameFormat = api_getVideoFrameFormat;
    a.getFrameWidth = api_getFrameWidth;
    a.getFrameHeight = api_getFrameHeight;
    a.getFrame = api_getFrame;
    a.getFrameFilter = api_getFrameFilter;
    a.requestFrameFilter = api_requestFrameFilter;
    a.setFilterError = api_setFilterError;
    a.createMap = api_createMap;
    a.freeMap = api_freeMap;
    a.mapSetError = api_mapSetError;
    a.mapGetError = api_mapGetError;
    a.mapNumElements = api_mapNumElements;
    a.mapGetType = api_mapGetType;
    a.mapGetInt = api_mapGetInt;
    a.mapSetInt = api_mapSetInt;
    a.mapGetFloat = api_mapGetFloat;
    a.mapSetFloat = api_mapSetFloat;
    a.mapGetData = api_mapGetData;
    a.mapGetDataSize = api_mapGetDataSize;
    a.mapSetData = api_mapSetData;
    a.mapGetNode = api_mapGetNode;
    a.mapSetNode = api_mapSetNode;
    a.mapConsumeNode = api_mapConsumeNode;
    return a;
}
const VSAPI g_api_table = make_api();

void VS_CC api_freeNode(VSNode* node) {
    if (!node || --node->refs > 0) return;
    if (node->free_fn) node->free_fn(node->instance, node->core, &g_api_table);
    for (VSFrame* f : node->frames) frame_unref(f);
    --node->core->live_nodes;
    delete node;
}

// Frame n of a node: source nodes hand out their stored frame; filter nodes run the two-step protocol.
const VSFrame* produce(VSNode* node, int n, std::string& error, const VSAPI* api) {
    if (!node->get_frame) {
        ++node->get_frame_calls;
        if (n < 0 || n >= static_cast<int>(node->frames.size())) {
            error = "source has no such frame";
            return nullptr;
        }
        ++node->frames[static_cast<size_t>(n)]->refs;
        return node->frames[static_cast<size_t>(n)];
    }
    VSFrameContext ctx;
    void* frame_data = nullptr;
    const VSFrame* out = node->get_frame(n, arInitial, node->instance, &frame_data, &ctx, node->core, api);
    if (!out && !ctx.has_error) out = node->get_frame(n, arAllFramesReady, node->instance, &frame_data, &ctx, node->core, api);
    for (auto& kv : ctx.ready) frame_unref(kv.second);
    if (ctx.has_error) {
        error = ctx.error;
        if (out) frame_unref(out);
        return nullptr;
    }
    return out;
}

const VSFrame* VS_CC api_getFrame(int n, VSNode* node, char* errorMsg, int bufSize) {
    std::string err;
    const VSFrame* f = produce(node, n, err, &g_api_table);
    if (!f && errorMsg && bufSize > 0) {
        std::strncpy(errorMsg, err.c_str(), static_cast<size_t>(bufSize) - 1);
        errorMsg[bufSize - 1] = '\0';
    }
    return f;
}

// ---- VSPLUGINAPI ----
int VS_CC papi_getAPIVersion(void) { return VAPOURSYNTH_API_VERSION; }
int VS_CC papi_configPlugin(const char* identifier, const char* ns, const char* name, int, int apiVersion, int, VSPlugin* plugin) {
    plugin->identifier = identifier;
    plugin->ns = ns;
    plugin->name = name;
    plugin->api_version = apiVersion;
    return 1;
}
int VS_CC papi_registerFunction(const char* name, const char* args, const char* ret, VSPublicFunction fn, void* data, VSPlugin* plugin) {
    plugin->fns.push_back({name, args, ret, fn, data});
    return 1;
}
const VSPLUGINAPI g_papi = {papi_getAPIVersion, papi_configPlugin, papi_registerFunction};

struct ArgSpec {
    std::string name, type;
    bool opt = false;
};
std::vector<ArgSpec> parse_args(const std::string& sig) {
    std::vector<ArgSpec> out;
    std::stringstream ss(sig);
    std::string item;
    while (std::getline(ss, item, ';')) {
        if (item.empty()) continue;
        std::stringstream is(item);
        std::string part;
        ArgSpec a;
        std::getline(is, a.name, ':');
        std::getline(is, a.type, ':');
        while (std::getline(is, part, ':'))
            if (part == "opt") a.opt = true;
        out.push_back(a);
    }
    return out;
}
}  // namespace

// ---- C interface for the Python test (ctypes) ----
extern "C" {
#define MOCK_API __attribute__((visibility("default")))

struct MockVs {
    VSCore core;
    VSPlugin plugin;
    std::string last_error;
};

MOCK_API MockVs* mockvs_new(int stride_align) {
    MockVs* m = new MockVs;
    m->core.stride_align = stride_align > 0 ? stride_align : 64;
    g_core_of_api = &m->core;
    g_api = &g_api_table;
    VapourSynthPluginInit2(&m->plugin, &g_papi);
    return m;
}
MOCK_API void mockvs_free(MockVs* m) { delete m; }
MOCK_API const char* mockvs_plugin_namespace(MockVs* m) { return m->plugin.ns.c_str(); }
MOCK_API const char* mockvs_plugin_identifier(MockVs* m) { return m->plugin.identifier.c_str(); }
MOCK_API int mockvs_plugin_api_version(MockVs* m) { return m->plugin.api_version; }
MOCK_API int mockvs_function_count(MockVs* m) { return static_cast<int>(m->plugin.fns.size()); }
MOCK_API const char* mockvs_function_name(MockVs* m, int i) { return m->plugin.fns[static_cast<size_t>(i)].name.c_str(); }
MOCK_API const char* mockvs_function_args(MockVs* m, int i) { return m->plugin.fns[static_cast<size_t>(i)].args.c_str(); }
MOCK_API const char* mockvs_function_return(MockVs* m, int i) { return m->plugin.fns[static_cast<size_t>(i)].ret.c_str(); }
MOCK_API long mockvs_live_frames(MockVs* m) { return m->core.live_frames; }
MOCK_API long mockvs_live_nodes(MockVs* m) { return m->core.live_nodes; }

// color_family: 1 gray, 2 RGB, 3 YUV; sample_type: 0 integer, 1 float
MOCK_API VSNode* mockvs_source_new(MockVs* m, int w, int h, int color_family, int sample_type, int bits, int bytes, int sub_w, int sub_h, int planes,
                                   int num_frames, int chroma_location) {
    VSNode* n = new VSNode;
    n->core = &m->core;
    n->vi.format = VSVideoFormat{color_family, sample_type, bits, bytes, sub_w, sub_h, planes};
    n->vi.fpsNum = 24;
    n->vi.fpsDen = 1;
    n->vi.width = w;
    n->vi.height = h;
    n->vi.numFrames = num_frames;
    ++m->core.live_nodes;
    for (int i = 0; i < num_frames; ++i) {
        VSFrame* f = new_frame(&m->core, n->vi.format, w, h);
        if (chroma_location >= 0) api_mapSetInt(&f->props, "_ChromaLocation", chroma_location, maReplace);
        api_mapSetInt(&f->props, "_MockFrameNumber", i, maReplace);  // an unrelated property: must survive the filter
        n->frames.push_back(f);
    }
    return n;
}
MOCK_API VSFrame* mockvs_source_frame(VSNode* n, int i) { return n->frames[static_cast<size_t>(i)]; }
MOCK_API int mockvs_source_get_frame_calls(VSNode* n) { return n->get_frame_calls; }
MOCK_API void mockvs_node_release(VSNode* n) { api_freeNode(n); }
MOCK_API uint8_t* mockvs_frame_plane(VSFrame* f, int plane, int* stride, int* row_bytes, int* height) {
    *stride = static_cast<int>(f->stride[plane]);
    *row_bytes = f->pw[plane] * f->format.bytesPerSample;
    *height = f->ph[plane];
    return f->buf[plane].data();
}
MOCK_API int mockvs_frame_prop_int(VSFrame* f, const char* key, long long* out) {
    auto it = f->props.kv.find(key);
    if (it == f->props.kv.end() || it->second.type != ptInt || it->second.ints.empty()) return 0;
    *out = it->second.ints[0];
    return 1;
}
MOCK_API void mockvs_frame_release(VSFrame* f) { frame_unref(f); }

// Calls jinc.<name>(clip, target_width, target_height, **named); kinds[i]: 'i' / 'f' / 's'.  Arguments are checked against
// the registered signature (unknown names, wrong types, missing required arguments) the way the core does before it calls
// the plugin.  Returns the output node or NULL (mockvs_last_error).
MOCK_API VSNode* mockvs_invoke(MockVs* m, const char* name, VSNode* clip, int tw, int th, int n, const char** names, const char* kinds,
                               const int* ivals, const double* fvals, const char** svals) {
    m->last_error.clear();
    const VSPlugin::Fn* fn = nullptr;
    for (const auto& f : m->plugin.fns)
        if (f.name == name) fn = &f;
    if (!fn) {
        m->last_error = std::string("no function ") + name;
        return nullptr;
    }
    VSMap in, out;
    api_mapSetNode(&in, "clip", clip, maReplace);
    api_mapSetInt(&in, "target_width", tw, maReplace);
    api_mapSetInt(&in, "target_height", th, maReplace);
    for (int i = 0; i < n; ++i) {
        if (kinds[i] == 'i') api_mapSetInt(&in, names[i], ivals[i], maReplace);
        else if (kinds[i] == 'f') api_mapSetFloat(&in, names[i], fvals[i], maReplace);
        else api_mapSetData(&in, names[i], svals[i], -1, dtUtf8, maReplace);
    }
    const std::vector<ArgSpec> spec = parse_args(fn->args);
    for (const auto& kv : in.kv) {
        const ArgSpec* s = nullptr;
        for (const auto& a : spec)
            if (a.name == kv.first) s = &a;
        if (!s) {
            m->last_error = std::string(name) + ": Function does not take argument(s) named " + kv.first;
            break;
        }
        const int want = s->type == "int" ? ptInt : s->type == "float" ? ptFloat : s->type == "data" ? ptData : s->type == "vnode" ? ptVideoNode : ptUnset;
        if (kv.second.type != want && !(want == ptFloat && kv.second.type == ptInt)) {
            m->last_error = std::string(name) + ": argument " + kv.first + " is not of the correct type";
            break;
        }
    }
    for (const auto& a : spec)
        if (!a.opt && !in.kv.count(a.name) && m->last_error.empty()) m->last_error = std::string(name) + ": argument " + a.name + " is required";
    VSNode* result = nullptr;
    if (m->last_error.empty()) {
        for (auto& kv : in.kv)  // the core converts ints given for float arguments
            for (const auto& a : spec)
                if (a.name == kv.first && a.type == "float" && kv.second.type == ptInt) {
                    const double d = static_cast<double>(kv.second.ints[0]);
                    kv.second = MapValue{};
                    kv.second.type = ptFloat;
                    kv.second.floats.push_back(d);
                }
        fn->fn(&in, &out, fn->data, &m->core, &g_api_table);
        if (out.has_error) m->last_error = out.error;
        else if (out.kv.count("clip") && !out.kv["clip"].nodes.empty()) result = api_addNodeRef(out.kv["clip"].nodes[0]);
        else m->last_error = "the function returned no clip";
    }
    release_map_nodes(&in);
    release_map_nodes(&out);
    return result;
}
MOCK_API const char* mockvs_last_error(MockVs* m) { return m->last_error.c_str(); }
MOCK_API void mockvs_node_info(VSNode* n, int* w, int* h, int* frames, int* filter_mode) {
    *w = n->vi.width;
    *h = n->vi.height;
    *frames = n->vi.numFrames;
    *filter_mode = n->filter_mode;
}
// Frame n of a node through the request protocol; NULL + message on failure.
MOCK_API VSFrame* mockvs_get_frame(MockVs* m, VSNode* node, int n) {
    m->last_error.clear();
    std::string err;
    const VSFrame* f = produce(node, n, err, &g_api_table);
    if (!f) m->last_error = err;
    return const_cast<VSFrame*>(f);
}
}  // extern "C"
