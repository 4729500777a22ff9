/* plugin/compat/VapourSynth4.h -- COMPATIBILITY DECLARATION of the part of the VapourSynth API 4 that
 * plugin/jincresize_vs.cpp uses.
 *
 * The VapourSynth SDK header is not part of this repository or its build image.  This file is written from the API's
 * public names and call shapes so that the VapourSynth front-end (the lineage the reference names in its
 * README.md:5) can be compiled here -- into plugin/lib/libjincresize_vs.so by __graft_entry__.build(), and together with
 * the mock host of tests/mock_vs/ for the plugin tests.  When building against a real VapourSynth installation, put the
 * SDK's include directory FIRST on the include path so that its VapourSynth4.h replaces this one.
 *
 * TO BE VERIFIED AGAINST UPSTREAM VapourSynth4.h before trusting a binary built with this file:
 *   - the ORDER of the function pointers in struct VSAPI and VSPLUGINAPI (binary compatibility rests on it; the order
 *     below is the upstream order as far as the author knows it -- every member up to the last one the plugin uses is
 *     declared, unused ones as generic pointers)
 *   - numeric values of the enums (VSColorFamily, VSSampleType, VSFilterMode, VSActivationReason, VSMapAppendMode,
 *     VSRequestPattern, VSPropertyType) and VAPOURSYNTH_API_MAJOR / _MINOR
 *   - field order of VSVideoFormat / VSVideoInfo / VSFilterDependency
 *   - VS_CC (stdcall on 32-bit Windows) and the VS_EXTERNAL_API export declaration of VapourSynthPluginInit2
 */
#ifndef JINCRESIZE_COMPAT_VAPOURSYNTH4_H
#define JINCRESIZE_COMPAT_VAPOURSYNTH4_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_CC
#define VS_EXTERNAL_API(ret) __attribute__((visibility("default"))) ret VS_CC

#define VAPOURSYNTH_API_MAJOR 4
#define VAPOURSYNTH_API_MINOR 0
#define VS_MAKE_VERSION(major, minor) (((major) << 16) | (minor))
#define VAPOURSYNTH_API_VERSION VS_MAKE_VERSION(VAPOURSYNTH_API_MAJOR, VAPOURSYNTH_API_MINOR)

typedef struct VSFrame VSFrame;
typedef struct VSNode VSNode;
typedef struct VSCore VSCore;
typedef struct VSPlugin VSPlugin;
typedef struct VSPluginFunction VSPluginFunction;
typedef struct VSFunction VSFunction;
typedef struct VSMap VSMap;
typedef struct VSLogHandle VSLogHandle;
typedef struct VSFrameContext VSFrameContext;
typedef struct VSPLUGINAPI VSPLUGINAPI;
typedef struct VSAPI VSAPI;

typedef enum VSColorFamily { cfUndefined = 0, cfGray = 1, cfRGB = 2, cfYUV = 3 } VSColorFamily;
typedef enum VSSampleType { stInteger = 0, stFloat = 1 } VSSampleType;
typedef enum VSFilterMode { fmParallel = 0, fmParallelRequests = 1, fmUnordered = 2, fmFrameState = 3 } VSFilterMode;
typedef enum VSActivationReason { arError = -1, arInitial = 0, arAllFramesReady = 1 } VSActivationReason;
typedef enum VSMapAppendMode { maReplace = 0, maAppend = 1 } VSMapAppendMode;
typedef enum VSRequestPattern { rpGeneral = 0, rpNoFrameReuse = 1, rpStrictSpatial = 2 } VSRequestPattern;
typedef enum VSPropertyType { ptUnset = 0, ptInt = 1, ptFloat = 2, ptData = 3, ptFunction = 4, ptVideoNode = 5, ptAudioNode = 6,
                              ptVideoFrame = 7, ptAudioFrame = 8 } VSPropertyType;
typedef enum VSDataTypeHint { dtUnknown = -1, dtBinary = 0, dtUtf8 = 1 } VSDataTypeHint;

typedef struct VSVideoFormat {
    int colorFamily;    /* VSColorFamily */
    int sampleType;     /* VSSampleType */
    int bitsPerSample;
    int bytesPerSample;
    int subSamplingW;   /* log2 */
    int subSamplingH;
    int numPlanes;
} VSVideoFormat;

typedef struct VSVideoInfo {
    VSVideoFormat format;
    int64_t fpsNum;
    int64_t fpsDen;
    int width;
    int height;
    int numFrames;
} VSVideoInfo;

typedef struct VSFilterDependency {
    VSNode *source;
    int requestPattern; /* VSRequestPattern */
} VSFilterDependency;

typedef void(VS_CC *VSPublicFunction)(const VSMap *in, VSMap *out, void *userData, VSCore *core, const VSAPI *vsapi);
typedef void(VS_CC *VSInitPlugin)(VSPlugin *plugin, const VSPLUGINAPI *vspapi);
typedef const VSFrame *(VS_CC *VSFilterGetFrame)(int n, int activationReason, void *instanceData, void **frameData,
                                                 VSFrameContext *frameCtx, VSCore *core, const VSAPI *vsapi);
typedef void(VS_CC *VSFilterFree)(void *instanceData, VSCore *core, const VSAPI *vsapi);

struct VSPLUGINAPI {
    int(VS_CC *getAPIVersion)(void);
    int(VS_CC *configPlugin)(const char *identifier, const char *pluginNamespace, const char *name, int pluginVersion, int apiVersion,
                             int flags, VSPlugin *plugin);
    int(VS_CC *registerFunction)(const char *name, const char *args, const char *returnType, VSPublicFunction argsFunc,
                                 void *functionData, VSPlugin *plugin);
};

typedef void (*VSUnusedFn)(void); /* members the plugin never calls, kept for their position */

struct VSAPI {
    /* filters and nodes */
    void(VS_CC *createVideoFilter)(VSMap *out, const char *name, const VSVideoInfo *vi, VSFilterGetFrame getFrame, VSFilterFree free,
                                   int filterMode, const VSFilterDependency *dependencies, int numDeps, void *instanceData, VSCore *core);
    VSUnusedFn createVideoFilter2;
    VSUnusedFn createAudioFilter;
    VSUnusedFn createAudioFilter2;
    VSUnusedFn setLinearFilter;
    VSUnusedFn setCacheMode;
    VSUnusedFn setCacheOptions;
    void(VS_CC *freeNode)(VSNode *node);
    VSNode *(VS_CC *addNodeRef)(VSNode *node);
    VSUnusedFn getNodeType;
    const VSVideoInfo *(VS_CC *getVideoInfo)(VSNode *node);
    VSUnusedFn getAudioInfo;
    /* frames */
    VSFrame *(VS_CC *newVideoFrame)(const VSVideoFormat *format, int width, int height, const VSFrame *propSrc, VSCore *core);
    VSUnusedFn newVideoFrame2;
    VSUnusedFn newAudioFrame;
    VSUnusedFn newAudioFrame2;
    void(VS_CC *freeFrame)(const VSFrame *f);
    VSUnusedFn addFrameRef;
    VSUnusedFn copyFrame;
    const VSMap *(VS_CC *getFramePropertiesRO)(const VSFrame *f);
    VSMap *(VS_CC *getFramePropertiesRW)(VSFrame *f);
    ptrdiff_t(VS_CC *getStride)(const VSFrame *f, int plane);
    const uint8_t *(VS_CC *getReadPtr)(const VSFrame *f, int plane);
    uint8_t *(VS_CC *getWritePtr)(VSFrame *f, int plane);
    const VSVideoFormat *(VS_CC *getVideoFrameFormat)(const VSFrame *f);
    VSUnusedFn getAudioFrameFormat;
    VSUnusedFn getFrameType;
    int(VS_CC *getFrameWidth)(const VSFrame *f, int plane);
    int(VS_CC *getFrameHeight)(const VSFrame *f, int plane);
    VSUnusedFn getFrameLength;
    /* formats */
    VSUnusedFn getVideoFormatName;
    VSUnusedFn getAudioFormatName;
    VSUnusedFn queryVideoFormat;
    VSUnusedFn queryAudioFormat;
    VSUnusedFn queryVideoFormatID;
    VSUnusedFn getVideoFormatByID;
    /* frame requests */
    const VSFrame *(VS_CC *getFrame)(int n, VSNode *node, char *errorMsg, int bufSize); /* synchronous; filter creation only */
    VSUnusedFn getFrameAsync;
    const VSFrame *(VS_CC *getFrameFilter)(int n, VSNode *node, VSFrameContext *frameCtx);
    void(VS_CC *requestFrameFilter)(int n, VSNode *node, VSFrameContext *frameCtx);
    VSUnusedFn releaseFrameEarly;
    VSUnusedFn cacheFrame;
    void(VS_CC *setFilterError)(const char *errorMessage, VSFrameContext *frameCtx);
    /* external functions */
    VSUnusedFn createFunction;
    VSUnusedFn freeFunction;
    VSUnusedFn addFunctionRef;
    VSUnusedFn callFunction;
    /* maps */
    VSMap *(VS_CC *createMap)(void);
    void(VS_CC *freeMap)(VSMap *map);
    VSUnusedFn clearMap;
    VSUnusedFn copyMap;
    void(VS_CC *mapSetError)(VSMap *map, const char *errorMessage);
    const char *(VS_CC *mapGetError)(const VSMap *map);
    VSUnusedFn mapNumKeys;
    VSUnusedFn mapGetKey;
    VSUnusedFn mapDeleteKey;
    int(VS_CC *mapNumElements)(const VSMap *map, const char *key); /* -1: key not present */
    int(VS_CC *mapGetType)(const VSMap *map, const char *key);
    VSUnusedFn mapSetEmpty;
    int64_t(VS_CC *mapGetInt)(const VSMap *map, const char *key, int index, int *error);
    VSUnusedFn mapGetIntSaturated;
    VSUnusedFn mapGetIntArray;
    int(VS_CC *mapSetInt)(VSMap *map, const char *key, int64_t i, int append);
    VSUnusedFn mapSetIntArray;
    double(VS_CC *mapGetFloat)(const VSMap *map, const char *key, int index, int *error);
    VSUnusedFn mapGetFloatSaturated;
    VSUnusedFn mapGetFloatArray;
    int(VS_CC *mapSetFloat)(VSMap *map, const char *key, double d, int append);
    VSUnusedFn mapSetFloatArray;
    const char *(VS_CC *mapGetData)(const VSMap *map, const char *key, int index, int *error);
    int(VS_CC *mapGetDataSize)(const VSMap *map, const char *key, int index, int *error);
    VSUnusedFn mapGetDataTypeHint;
    int(VS_CC *mapSetData)(VSMap *map, const char *key, const char *data, int size, int type, int append);
    VSNode *(VS_CC *mapGetNode)(const VSMap *map, const char *key, int index, int *error);
    int(VS_CC *mapSetNode)(VSMap *map, const char *key, VSNode *node, int append);
    int(VS_CC *mapConsumeNode)(VSMap *map, const char *key, VSNode *node, int append);
    /* (frames and functions in maps, plugin enumeration, invoke, core functions and logging follow in the upstream header; the
     * plugin uses none of them) */
};

VS_EXTERNAL_API(void) VapourSynthPluginInit2(VSPlugin *plugin, const VSPLUGINAPI *vspapi);

#ifdef __cplusplus
}
#endif
#endif /* JINCRESIZE_COMPAT_VAPOURSYNTH4_H */
