// scene_cache.cpp — the reference's scene cache in its JSON multi-file form (BVH_ReSTIR_DL/Assets/Scripts/FileLoader.cs:12-91,
// Data types/Other/{MultiArrayWrapper,Wrapper}.cs): one `<key>.json` per array, each `{ "array": [ ... ] }`, structs
// as objects of their public fields, matrices as columns c0..c3 of x,y,z,w (Unity.Mathematics float4x4), floats
// printed as the shortest double that round-trips.
//
// Read: what ObjectHelper.ConstructScene returns in LoadExistingFile mode (ObjectHelper.cs:672-677,493-517) —
// loadedBVs, loadedVertices, renderTriangles, sceneObjectDatas, lightObjects plus the keyed integers and floats —
// and loadedComponentDatas.  Written: all twelve files of SaveMultiArrayContainerToJsonFiles (FileLoader.cs:26-44), so that
// LoadMultiArrayContainerFromJsonFiles (FileLoader.cs:63-91) finds every array it wraps: the ones above, the empty
// renderBVs.json the reference's own writer leaves, and the three that only the reference's scene construction reads
// back (ObjectHelper.cs:499-504,627-631) - loadedTriangles (Triangle.cs:7-13: indices, parentIndex, min, max, area),
// loadedMeshesLookup (scene object -> mesh) and loadedMeshes (MeshData.cs:6-11).  A mesh is a distinct bvStartIndex;
// its triangles are the root volume's component range, its local box the root volume's.  meshKey is Unity's digest of
// the imported Mesh asset (Resources.cs:90-135), which a scene built here does not have: the keys written are
// "rth-mesh-<n>", which match no asset, so Unity re-imports a mesh it is asked to add again instead of mistaking it.
// Files may carry a `<prefix>_` in front of the key, as the copy committed with the reference does (`80kScene_`).
#include "rth_api.h"
#include "scene_internal.h"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

struct ParseError { std::string msg; };

// Minimal JSON reader for this format: calls `onElement` once per element of the top-level "array", handing it the
// element flattened to path -> number ("localToWorldMatrix.c0.x"), path -> string for string leaves, and the bare
// number for scalar elements (path "").
struct Flat {
    std::map<std::string, double> num;
    std::map<std::string, std::string> str;
    double get(const char* k) const {
        auto it = num.find(k);
        if (it == num.end()) throw ParseError{std::string("missing field '") + k + "'"};
        return it->second;
    }
    double getOr(const char* k, double d) const { auto it = num.find(k); return it == num.end() ? d : it->second; }
};
struct Reader {
    const char* p;
    const char* end;
    explicit Reader(const std::string& s) : p(s.data()), end(s.data() + s.size()) {}
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; }
    bool eat(char c) { ws(); if (p < end && *p == c) { p++; return true; } return false; }
    void need(char c) { if (!eat(c)) throw ParseError{std::string("expected '") + c + "' " + std::to_string((long)(end - p)) + " bytes before the end"}; }
    std::string string() {
        need('"');
        std::string out;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) { p++; out.push_back(*p == 'n' ? '\n' : *p == 't' ? '\t' : *p); p++; }
            else out.push_back(*p++);
        }
        if (p >= end) throw ParseError{"unterminated string"};
        p++;
        return out;
    }
    double number() {
        ws();
        char* e = nullptr;
        errno = 0;
        double v = strtod(p, &e);
        if (e == p) throw ParseError{"expected a number"};
        p = e;
        return v;
    }
    void value(const std::string& path, Flat& f) {
        ws();
        if (p >= end) throw ParseError{"unexpected end of file"};
        if (*p == '{') {
            p++;
            if (eat('}')) return;
            do {
                std::string k = string();
                need(':');
                value(path.empty() ? k : path + "." + k, f);
            } while (eat(','));
            need('}');
        } else if (*p == '[') {  // nested arrays do not occur in the cached types; skip them structurally
            p++;
            if (eat(']')) return;
            int i = 0;
            do { value(path + "[" + std::to_string(i++) + "]", f); } while (eat(','));
            need(']');
        } else if (*p == '"') {
            f.str[path] = string();
        } else if (!strncmp(p, "true", 4)) { f.num[path] = 1; p += 4; }
        else if (!strncmp(p, "false", 5)) { f.num[path] = 0; p += 5; }
        else if (!strncmp(p, "null", 4)) { p += 4; }
        else f.num[path] = number();
    }
    template <class Fn>
    void array(Fn onElement) {
        need('{');
        std::string k = string();
        if (k != "array") throw ParseError{"top-level key is '" + k + "', expected 'array'"};
        need(':');
        need('[');
        if (eat(']')) { need('}'); return; }
        do {
            Flat f;
            value("", f);
            onElement(f);
        } while (eat(','));
        need(']');
        need('}');
    }
};

bool readFile(const std::string& path, std::string& out) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    out.resize((size_t)(n > 0 ? n : 0));
    size_t got = n > 0 ? fread(&out[0], 1, (size_t)n, fp) : 0;
    fclose(fp);
    return got == (size_t)(n > 0 ? n : 0);
}
// `<folder>/<key>.json`, or `<folder>/<prefix>_<key>.json`.
bool loadKey(const std::string& folder, const std::string& prefix, const char* key, std::string& text) {
    if (readFile(folder + "/" + key + ".json", text)) return true;
    return !prefix.empty() && readFile(folder + "/" + prefix + "_" + key + ".json", text);
}

rt_float3 f3(const Flat& f, const std::string& k) {
    return {(float)f.get((k + ".x").c_str()), (float)f.get((k + ".y").c_str()), (float)f.get((k + ".z").c_str())};
}
rt_float4x4 m44(const Flat& f, const std::string& k) {
    rt_float4x4 m;
    const char* comp[4] = {"x", "y", "z", "w"};
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) m.m[c * 4 + r] = (float)f.get((k + ".c" + std::to_string(c) + "." + comp[r]).c_str());
    return m;
}

// ---- writer ----
struct Writer {
    FILE* fp;
    bool first = true;
    explicit Writer(FILE* f) : fp(f) { fputs("{\n    \"array\": [", fp); }
    void sep() { fputs(first ? "\n" : ",\n", fp); first = false; }
    void close() { fputs(first ? "]\n}" : "\n    ]\n}", fp); }
};
// A float the way JsonUtility prints it: as a double, shortest digits that round-trip, integers with ".0".
std::string num(float v) {
    double d = (double)v;
    if (std::isnan(d)) return "NaN";
    if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
    char buf[40];
    for (int prec = 1; prec <= 17; prec++) {
        snprintf(buf, sizeof buf, "%.*g", prec, d);
        if (strtod(buf, nullptr) == d) break;
    }
    std::string s = buf;
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
    return s;
}
std::string vec3(const rt_float3& v, const char* ind) {
    return std::string("{\n") + ind + "    \"x\": " + num(v.x) + ",\n" + ind + "    \"y\": " + num(v.y) + ",\n" + ind + "    \"z\": " + num(v.z) + "\n" + ind + "}";
}
std::string mat(const rt_float4x4& m, const char* ind) {
    std::string s = "{\n";
    const char* comp[4] = {"x", "y", "z", "w"};
    for (int c = 0; c < 4; c++) {
        s += std::string(ind) + "    \"c" + std::to_string(c) + "\": {\n";
        for (int r = 0; r < 4; r++)
            s += std::string(ind) + "        \"" + comp[r] + "\": " + num(m.m[c * 4 + r]) + (r < 3 ? ",\n" : "\n");
        s += std::string(ind) + "    }" + (c < 3 ? ",\n" : "\n");
    }
    return s + ind + "}";
}
const char* I3 = "            ";  // indentation of a field inside an array element

template <class Fn>
bool writeKey(const std::string& folder, const char* key, Fn body) {
    FILE* fp = fopen((folder + "/" + key + ".json").c_str(), "wb");
    if (!fp) return false;
    Writer w(fp);
    body(w);
    w.close();
    return fclose(fp) == 0;
}

}  // namespace

extern "C" {

int rth_scene_save_json(rth_scene s, const char* folder) {
    rth_scene_data d;
    if (!s || !folder || rth_scene_get(s, &d) != 0) return -1;
    const std::string dir = folder;
    bool ok = true;
    ok &= writeKey(dir, "loadedBVs", [&](Writer& w) {
        for (int64_t i = 0; i < d.nBVs; i++) {
            w.sep();
            fprintf(w.fp, "        {\n%s\"min\": %s,\n%s\"max\": %s,\n%s\"indexA\": %d,\n%s\"indexB\": %d\n        }", I3, vec3(d.BVs[i].min, I3).c_str(),
                    I3, vec3(d.BVs[i].max, I3).c_str(), I3, d.BVs[i].indexA, I3, d.BVs[i].indexB);
        }
    });
    ok &= writeKey(dir, "renderBVs", [&](Writer&) {});  // never filled by SaveToFile (ObjectHelper.cs:523-544)
    ok &= writeKey(dir, "loadedVertices", [&](Writer& w) {
        for (int64_t i = 0; i < d.nVertices; i++) {
            w.sep();
            fprintf(w.fp, "        {\n%s\"pos\": %s,\n%s\"uv\": {\n%s    \"x\": %s,\n%s    \"y\": %s\n%s}\n        }", I3, vec3(d.Vertices[i].pos, I3).c_str(), I3, I3,
                    num(d.Vertices[i].uv.x).c_str(), I3, num(d.Vertices[i].uv.y).c_str(), I3);
        }
    });
    ok &= writeKey(dir, "renderTriangles", [&](Writer& w) {
        for (int64_t i = 0; i < d.nTriangles; i++) {
            const rt_triangle& t = d.Triangles[i];
            w.sep();
            fprintf(w.fp, "        {\n%s\"vertex0Index\": %d,\n%s\"vertex1Index\": %d,\n%s\"vertex2Index\": %d,\n%s\"localNormal\": %s,\n%s\"area\": %s\n        }", I3,
                    t.vertex0Index, I3, t.vertex1Index, I3, t.vertex2Index, I3, vec3(t.localNormal, I3).c_str(), I3, num(t.area).c_str());
        }
    });
    ok &= writeKey(dir, "sceneObjectDatas", [&](Writer& w) {
        for (int64_t i = 0; i < d.nSceneObjects; i++) {
            const rt_scene_object& o = d.SceneObjects[i];
            w.sep();
            fprintf(w.fp, "        {\n%s\"worldToLocalMatrix\": %s,\n%s\"localToWorldMatrix\": %s,\n%s\"materialIndex\": %d,\n%s\"bvStartIndex\": %d,\n%s\"maxDepthBVH\": %d,\n"
                          "%s\"areaApprox\": %s,\n%s\"min\": %s,\n%s\"max\": %s\n        }",
                    I3, mat(o.worldToLocalMatrix, I3).c_str(), I3, mat(o.localToWorldMatrix, I3).c_str(), I3, o.materialIndex, I3, o.bvStartIndex, I3,
                    o.maxDepthBVH, I3, num(o.areaApprox).c_str(), I3, vec3(o.min, I3).c_str(), I3, vec3(o.max, I3).c_str());
        }
    });
    ok &= writeKey(dir, "lightObjects", [&](Writer& w) {
        for (int64_t i = 0; i < d.nLightObjects; i++) {
            const rt_light_object& l = d.LightObjects[i];
            w.sep();
            fprintf(w.fp, "        {\n%s\"localToWorldMatrix\": %s,\n%s\"areaApprox\": %s,\n%s\"brightness\": %s,\n%s\"triStart\": %d,\n%s\"totTris\": %d\n        }", I3,
                    mat(l.localToWorldMatrix, I3).c_str(), I3, num(l.areaApprox).c_str(), I3, num(l.brightness).c_str(), I3, l.triStart, I3, l.totTris);
        }
    });
    ok &= writeKey(dir, "loadedComponentDatas", [&](Writer& w) {
        for (int64_t i = 0; i < d.nComponentDatas; i++) {
            w.sep();
            fprintf(w.fp, "        {\n%s\"x\": %d,\n%s\"y\": %d\n        }", I3, d.ComponentDatas[i].x, I3, d.ComponentDatas[i].y);
        }
    });
    // meshes: the distinct mesh BVH roots, in order of first use
    std::vector<int> meshRoot, lookup((size_t)d.nSceneObjects, 0);
    for (int64_t i = 0; i < d.nSceneObjects; i++) {
        const int root = d.SceneObjects[i].bvStartIndex;
        size_t m = 0;
        while (m < meshRoot.size() && meshRoot[m] != root) m++;
        if (m == meshRoot.size()) meshRoot.push_back(root);
        lookup[(size_t)i] = (int)m;
    }
    auto triFields = [&](const rt_triangle& t, const std::string& ind) {
        rt_float3 lo = d.Vertices[t.vertex0Index].pos, hi = lo;
        for (int vi : {t.vertex1Index, t.vertex2Index}) {
            const rt_float3 p = d.Vertices[vi].pos;
            lo = {std::fmin(lo.x, p.x), std::fmin(lo.y, p.y), std::fmin(lo.z, p.z)};
            hi = {std::fmax(hi.x, p.x), std::fmax(hi.y, p.y), std::fmax(hi.z, p.z)};
        }
        const char* in = ind.c_str();
        return ind + "\"vertex0Index\": " + std::to_string(t.vertex0Index) + ",\n" + ind + "\"vertex1Index\": " + std::to_string(t.vertex1Index) + ",\n" +
               ind + "\"vertex2Index\": " + std::to_string(t.vertex2Index) + ",\n" + ind + "\"parentIndex\": 0,\n" +   // never assigned by the reference
               ind + "\"min\": " + vec3(lo, in) + ",\n" + ind + "\"max\": " + vec3(hi, in) + ",\n" + ind + "\"area\": " + num(t.area) + "\n";
    };
    bool indexed = true;   // every triangle's vertices exist, every mesh root has its component entry
    for (int64_t i = 0; i < d.nTriangles && indexed; i++)
        for (int vi : {d.Triangles[i].vertex0Index, d.Triangles[i].vertex1Index, d.Triangles[i].vertex2Index}) indexed &= vi >= 0 && vi < d.nVertices;
    for (int root : meshRoot) {
        indexed &= root >= 0 && root < d.nComponentDatas && root < d.nBVs;
        if (indexed) indexed &= d.ComponentDatas[root].x >= 0 && d.ComponentDatas[root].y >= 0 && (int64_t)d.ComponentDatas[root].x + d.ComponentDatas[root].y <= d.nTriangles;
    }
    if (!indexed) return -1;
    ok &= writeKey(dir, "loadedTriangles", [&](Writer& w) {
        for (int64_t i = 0; i < d.nTriangles; i++) {
            w.sep();
            fprintf(w.fp, "        {\n%s        }", triFields(d.Triangles[i], I3).c_str());
        }
    });
    ok &= writeKey(dir, "loadedMeshesLookup", [&](Writer& w) {
        for (int m : lookup) { w.sep(); fprintf(w.fp, "        %d", m); }
    });
    ok &= writeKey(dir, "loadedMeshes", [&](Writer& w) {
        const std::string I5 = std::string(I3) + "        ";
        for (size_t m = 0; m < meshRoot.size(); m++) {
            const int root = meshRoot[m], start = d.ComponentDatas[root].x, count = d.ComponentDatas[root].y;
            w.sep();
            fprintf(w.fp, "        {\n%s\"triangles\": [", I3);
            for (int t = 0; t < count; t++)
                fprintf(w.fp, "%s\n%s    {\n%s%s    }", t ? "," : "", I3, triFields(d.Triangles[start + t], I5).c_str(), I3);
            fprintf(w.fp, "%s],\n%s\"localMin\": %s,\n%s\"localMax\": %s,\n%s\"componentStartIndex\": %d,\n%s\"bvStartIndex\": %d,\n%s\"meshKey\": \"rth-mesh-%zu\"\n        }",
                    count ? (std::string("\n") + I3).c_str() : "", I3, vec3(d.BVs[root].min, I3).c_str(), I3, vec3(d.BVs[root].max, I3).c_str(), I3, start, I3, root, I3, m);
        }
    });
    ok &= writeKey(dir, "integers", [&](Writer& w) {
        const char* keys[3] = {"maxBVHDepth", "emittingObjectsNum", "sceneBVHStartIndex"};
        const int vals[3] = {d.maxBVHDepth, d.emittingObjectsNum, d.sceneBVHStartIndex};
        for (int i = 0; i < 3; i++) { w.sep(); fprintf(w.fp, "        {\n%s\"Key\": \"%s\",\n%s\"Value\": %d\n        }", I3, keys[i], I3, vals[i]); }
    });
    ok &= writeKey(dir, "floats", [&](Writer& w) {
        w.sep();
        fprintf(w.fp, "        {\n%s\"Key\": \"totArea\",\n%s\"Value\": %s\n        }", I3, I3, num(d.totArea).c_str());
    });
    return ok ? 0 : -1;
}

int rth_scene_load_json(rth_scene s, const char* folder, const char* prefix, char* err, int errLen) {
    auto fail = [&](const std::string& m) { if (err && errLen > 0) snprintf(err, (size_t)errLen, "%s", m.c_str()); return -1; };
    if (!s || !folder) return fail("null argument");
    const std::string dir = folder, pre = prefix ? prefix : "";
    SceneArraysRef a = rth_scene_arrays_ref(s);
    std::string text;
    const char* key = "";
    try {
        auto open = [&](const char* k, bool required) -> bool {
            key = k;
            if (loadKey(dir, pre, k, text)) return true;
            if (required) throw ParseError{"file not found"};
            return false;
        };
        std::vector<rt_bv> BVs;
        open("loadedBVs", true);
        Reader(text).array([&](const Flat& f) {
            rt_bv b;
            b.min = f3(f, "min"); b.max = f3(f, "max"); b.indexA = (int)f.get("indexA"); b.indexB = (int)f.get("indexB");
            BVs.push_back(b);
        });
        std::vector<rt_vertex> verts;
        open("loadedVertices", true);
        Reader(text).array([&](const Flat& f) {
            rt_vertex v;
            v.pos = f3(f, "pos"); v.uv = {(float)f.get("uv.x"), (float)f.get("uv.y")};
            verts.push_back(v);
        });
        std::vector<rt_triangle> tris;
        open("renderTriangles", true);
        Reader(text).array([&](const Flat& f) {
            rt_triangle t;
            t.vertex0Index = (int)f.get("vertex0Index"); t.vertex1Index = (int)f.get("vertex1Index"); t.vertex2Index = (int)f.get("vertex2Index");
            t.localNormal = f3(f, "localNormal");
            t.area = (float)f.getOr("area", 0.0);  // absent from files written before the field was serialised
            tris.push_back(t);
        });
        std::vector<rt_scene_object> objs;
        open("sceneObjectDatas", true);
        Reader(text).array([&](const Flat& f) {
            rt_scene_object o;
            o.worldToLocalMatrix = m44(f, "worldToLocalMatrix"); o.localToWorldMatrix = m44(f, "localToWorldMatrix");
            o.materialIndex = (int)f.get("materialIndex"); o.bvStartIndex = (int)f.get("bvStartIndex"); o.maxDepthBVH = (int)f.get("maxDepthBVH");
            o.areaApprox = (float)f.get("areaApprox"); o.min = f3(f, "min"); o.max = f3(f, "max");
            objs.push_back(o);
        });
        std::vector<rt_light_object> lights;
        open("lightObjects", true);
        Reader(text).array([&](const Flat& f) {
            rt_light_object l;
            l.localToWorldMatrix = m44(f, "localToWorldMatrix"); l.areaApprox = (float)f.get("areaApprox"); l.brightness = (float)f.get("brightness");
            l.triStart = (int)f.get("triStart"); l.totTris = (int)f.get("totTris");
            lights.push_back(l);
        });
        std::vector<rt_int2> comps;
        if (open("loadedComponentDatas", false))
            Reader(text).array([&](const Flat& f) { comps.push_back({(int)f.get("x"), (int)f.get("y")}); });
        std::map<std::string, double> keyed;
        open("integers", true);
        Reader(text).array([&](const Flat& f) { auto k = f.str.find("Key"); if (k != f.str.end()) keyed[k->second] = f.get("Value"); });
        open("floats", true);
        Reader(text).array([&](const Flat& f) { auto k = f.str.find("Key"); if (k != f.str.end()) keyed[k->second] = f.get("Value"); });
        key = "integers/floats";
        for (const char* k : {"maxBVHDepth", "emittingObjectsNum", "sceneBVHStartIndex", "totArea"})
            if (!keyed.count(k)) throw ParseError{std::string("Key '") + k + "' not found"};  // GetIntByKey / GetFloatByKey
        *a.BVs = BVs; *a.vertices = verts; *a.triangles = tris; *a.sceneObjects = objs; *a.lightObjects = lights; *a.componentDatas = comps;
        *a.maxBVHDepth = (int)keyed["maxBVHDepth"]; *a.emittingObjectsNum = (int)keyed["emittingObjectsNum"];
        *a.sceneBVHStartIndex = (int)keyed["sceneBVHStartIndex"]; *a.totArea = (float)keyed["totArea"];
        if (!rth_scene_make_materials(s)) throw ParseError{"the scene's material textures cannot be packed"};
        *a.built = true;
    } catch (const ParseError& e) {
        return fail(std::string(key) + ".json: " + e.msg);
    }
    return 0;
}

}  // extern "C"
