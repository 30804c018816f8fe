// mipt_render -- headless renderer: what main.cpp + CS488Window::start
// (reference cpu/src/main.cpp:37-103, cs488.h:2598-2651) do, minus the
// OpenGL window: load .obj/.mtl (+ environment), build the SAH-BVH on the
// host, upload the arena, render on the GPU, write the image.
//
//   mipt_render --scene cornellbox.obj [--env probe.hdr] [--width 512 --height 384]
//               [--add-scene more.obj]... further objects of the same scene (Scene::addObject, cs488.h:2020), each with
//                                        its own BVH, in this order (mipt_scene_create_multi)
//               [--mode primary|direct|path] [--spp 16] [--bounces 4] [--seed 1]
//               [--eye x y z] [--lookat x y z] [--light x y z w] [--device 0]
//               [--out image.png] [--out-hdr image.hdr] [--out-pfm image.pfm]
//               [--traversal default|walk|bittrail] [--bvh reference|all-axes]
//               [--gpus N]               the frame's tiles over GPUs 0..N-1, gathered to GPU 0 with one RCCL collective
//                                        (mipt_node_*); the image is bit-identical to the one-GPU image
//               [--cache scene.miptsc]   mesh + BVH from this flat file when it exists and loads, else parsed and
//                                        built as usual and then written to it (mipt_scene_cache_*); the hierarchy
//                                        the traversal walks is kept beside it (<file>.wide, mipt_scene_create_cached)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/mipt.h"

static void die(const char* what) {
    fprintf(stderr, "mipt_render: %s: %s\n", what, mipt_last_error());
    exit(1);
}

int main(int argc, char** argv) {
    std::string scenePath, envPath, outPng, outHdr, outPfm, cachePath;
    std::vector<std::string> morePaths;
    mipt_camera cam{};
    cam.width = 512; cam.height = 384;  // globalWidth / globalHeight, cs488.h:45-46
    cam.eye[2] = 1.5f;                  // cs488.h:81-83
    cam.up[1] = 1.0f;
    cam.fov_deg = 45.0f; cam.film_size = 0.032f;
    mipt_render_params rp{};
    rp.mode = MIPT_MODE_DIRECT;  // the reference's default frame is Raytrace(), cs488.h:2620
    rp.spp = 1; rp.max_bounces = 4; rp.seed = 1; rp.part_count = 1;
    mipt_light light{{3.0f, 3.0f, 3.0f}, {1000.0f, 1000.0f, 1000.0f}};  // main.cpp:26-27
    int device = 0, traversal = MIPT_TRAVERSAL_DEFAULT, gpus = 0;
    bool allAxes = false;
    for (int i = 1; i < argc; i++) {
        auto need = [&](int k) { if (i + k >= argc) { fprintf(stderr, "missing value after %s\n", argv[i]); exit(2); } };
        std::string a = argv[i];
        if (a == "--scene") { need(1); scenePath = argv[++i]; }
        else if (a == "--add-scene") { need(1); morePaths.push_back(argv[++i]); }
        else if (a == "--env") { need(1); envPath = argv[++i]; }
        else if (a == "--width") { need(1); cam.width = atoi(argv[++i]); }
        else if (a == "--height") { need(1); cam.height = atoi(argv[++i]); }
        else if (a == "--spp") { need(1); rp.spp = atoi(argv[++i]); }
        else if (a == "--bounces") { need(1); rp.max_bounces = atoi(argv[++i]); }
        else if (a == "--seed") { need(1); rp.seed = (uint32_t)strtoul(argv[++i], nullptr, 10); }
        else if (a == "--device") { need(1); device = atoi(argv[++i]); }
        else if (a == "--gpus") { need(1); gpus = atoi(argv[++i]); }
        else if (a == "--out") { need(1); outPng = argv[++i]; }
        else if (a == "--out-hdr") { need(1); outHdr = argv[++i]; }
        else if (a == "--out-pfm") { need(1); outPfm = argv[++i]; }
        else if (a == "--cache") { need(1); cachePath = argv[++i]; }
        else if (a == "--traversal") {
            need(1);
            std::string t = argv[++i];
            traversal = t == "walk" ? MIPT_TRAVERSAL_THREADED : t == "bittrail" ? MIPT_TRAVERSAL_BITTRAIL : MIPT_TRAVERSAL_DEFAULT;
        } else if (a == "--bvh") { need(1); allAxes = std::string(argv[++i]) == "all-axes"; }
        else if (a == "--mode") {
            need(1);
            std::string m = argv[++i];
            rp.mode = m == "primary" ? MIPT_MODE_PRIMARY : m == "path" ? MIPT_MODE_PATH : MIPT_MODE_DIRECT;
        } else if (a == "--eye") { need(3); for (int k = 0; k < 3; k++) cam.eye[k] = (float)atof(argv[++i]); }
        else if (a == "--lookat") { need(3); for (int k = 0; k < 3; k++) cam.lookat[k] = (float)atof(argv[++i]); }
        else if (a == "--light") {
            need(4);
            for (int k = 0; k < 3; k++) light.position[k] = (float)atof(argv[++i]);
            const float w = (float)atof(argv[++i]);
            light.wattage[0] = light.wattage[1] = light.wattage[2] = w;
        } else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    mipt_mesh* mesh = nullptr;
    mipt_bvh* bvh = nullptr;
    auto t0 = std::chrono::high_resolution_clock::now();
    // what a cache must have been built from to stand in for this run: the .obj (and the .mtl beside it) by size and
    // content hash, and the BVH mode.  A cache with another tag is rebuilt and overwritten, not rendered.
    std::string tag;
    if (!cachePath.empty()) {
        char fp[512];
        mipt_file_fingerprint(scenePath.c_str(), fp, sizeof fp);
        tag = std::string("obj ") + fp;
        const size_t dot = scenePath.rfind('.');
        const std::string mtl = (dot == std::string::npos ? scenePath : scenePath.substr(0, dot)) + ".mtl";
        mipt_file_fingerprint(mtl.c_str(), fp, sizeof fp);
        tag += std::string("; mtl ") + fp + "; bvh " + (allAxes ? "all-axes" : "reference");
    }
    const bool cached = !cachePath.empty() && mipt_scene_cache_load_tagged(cachePath.c_str(), tag.c_str(), &mesh, &bvh) == MIPT_OK;
    if (!cachePath.empty() && !cached) printf("Cache not used: %s\n", mipt_last_error());
    if (cached) {
        printf("Loaded cache \"%s\" with %d triangles (%.1f ms).\n", cachePath.c_str(), mipt_mesh_num_triangles(mesh),
               std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count());
    } else {
        if (scenePath.empty() || mipt_mesh_load_obj(scenePath.c_str(), &mesh) != MIPT_OK) {
            // main.cpp:40-44: fall back to the single triangle when the .obj does not load
            fprintf(stderr, "Invalid .obj file (%s).\nMaking a single triangle instead.\n", mipt_last_error());
            if (mipt_mesh_create_single_triangle(&mesh) != MIPT_OK) die("single triangle");
        }
        printf("Loaded \"%s\" with %d triangles (%.1f ms).\n", scenePath.c_str(), mipt_mesh_num_triangles(mesh),
               std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count());
    }
    mipt_image* env = nullptr;
    if (!envPath.empty() && mipt_image_load(envPath.c_str(), &env) != MIPT_OK) die("environment");
    if (!cached) {
        t0 = std::chrono::high_resolution_clock::now();
        if ((allAxes ? mipt_bvh_build_all_axes(mesh, &bvh) : mipt_bvh_build(mesh, &bvh)) != MIPT_OK) die("BVH build");
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("Built BVH: %d nodes, %d leaves, depth %d (%.1f ms)\n", mipt_bvh_num_nodes(bvh), mipt_bvh_num_leaves(bvh),
               mipt_bvh_depth(bvh), std::chrono::duration<double, std::milli>(t1 - t0).count());
        if (!cachePath.empty() && mipt_scene_cache_save_tagged(cachePath.c_str(), mesh, bvh, tag.c_str()) != MIPT_OK)
            fprintf(stderr, "mipt_render: cache not written: %s\n", mipt_last_error());
    }
    mipt_scene_desc desc;
    if (mipt_scene_desc_fill(mesh, bvh, env, &light, 1, &desc) != MIPT_OK) die("scene description");
    mipt_scene* scene = nullptr;
    {
        // with --cache the library's own hierarchy above the leaves lives beside the flat scene file (<cache>.wide, keyed on
        // the tree, the leaf order, the positions and the light: mipt_scene_create_cached), so a second start builds nothing
        const auto t0 = std::chrono::steady_clock::now();
        const std::string widePath = cachePath.empty() ? std::string() : cachePath + ".wide";
        if (morePaths.empty()) {
            if (mipt_scene_create_cached(&desc, widePath.c_str(), &scene) != MIPT_OK) die("scene");
        } else {
            std::vector<mipt_scene_desc> descs{desc};
            for (const std::string& path : morePaths) {  // parsed and built every time: the flat cache holds the first object
                mipt_mesh* m = nullptr;
                mipt_bvh* b = nullptr;
                if (mipt_mesh_load_obj(path.c_str(), &m) != MIPT_OK) die(path.c_str());
                if ((allAxes ? mipt_bvh_build_all_axes(m, &b) : mipt_bvh_build(m, &b)) != MIPT_OK) die("BVH build");
                mipt_scene_desc d;
                if (mipt_scene_desc_fill(m, b, env, &light, 1, &d) != MIPT_OK) die("scene description");
                printf("Added \"%s\" with %d triangles.\n", path.c_str(), mipt_mesh_num_triangles(m));
                descs.push_back(d);  // m and b stay alive until the process ends
            }
            if (mipt_scene_create_multi(descs.data(), (int32_t)descs.size(), widePath.c_str(), &scene) != MIPT_OK) die("scene");
        }
        printf("Scene ready in %.1f ms (%d wide nodes%s).\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
               mipt_scene_wide_nodes(scene), widePath.empty() ? "" : ", hierarchy cache beside the scene cache");
    }
    if (cam.width <= 0 || cam.height <= 0 || cam.width > 32768 || cam.height > 32768) {
        fprintf(stderr, "mipt_render: --width / --height must be in 1..32768\n");
        return 2;
    }
    std::vector<float> rgb((size_t)3 * cam.width * cam.height);
    mipt_ctx* ctx = nullptr;
    mipt_node* node = nullptr;
    mipt_stats st;
    double ms;
    if (gpus > 0) {
        if (traversal != MIPT_TRAVERSAL_DEFAULT) { fprintf(stderr, "mipt_render: --gpus renders with the default traversal\n"); return 2; }
        if (mipt_node_create(scene, gpus, nullptr, &node) != MIPT_OK) die("node");
        printf("Node: %d GPU(s), gather over %s\n", mipt_node_num_gpus(node), mipt_node_uses_rccl(node) ? "RCCL" : "device copies");
        auto t2 = std::chrono::high_resolution_clock::now();
        if (mipt_node_render(node, &cam, &rp, rgb.data()) != MIPT_OK) die("render");
        ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t2).count();
        if (mipt_node_get_stats(node, &st, nullptr) != MIPT_OK) die("stats");
    } else {
        if (mipt_context_create(scene, device, nullptr, &ctx) != MIPT_OK) die("context");
        printf("Arena: %zu bytes on device %d\n", mipt_context_arena_bytes(ctx), device);
        if (mipt_context_set_traversal(ctx, traversal) != MIPT_OK) die("traversal");
        auto t2 = std::chrono::high_resolution_clock::now();
        if (mipt_render(ctx, &cam, &rp, rgb.data()) != MIPT_OK) die("render");
        ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t2).count();
        if (mipt_context_get_stats(ctx, &st) != MIPT_OK) die("stats");
    }
    printf("Frame duration %.2f ms, %llu rays (%.1f Mrays/s)\n", ms,
           (unsigned long long)(st.rays_closest + st.rays_occluded), (st.rays_closest + st.rays_occluded) / ms * 1e-3);
    if (!outPng.empty() && mipt_image_save_png(outPng.c_str(), cam.width, cam.height, rgb.data()) != MIPT_OK) die("save png");
    if (!outHdr.empty() && mipt_image_save_hdr(outHdr.c_str(), cam.width, cam.height, rgb.data()) != MIPT_OK) die("save hdr");
    if (!outPfm.empty() && mipt_image_save_pfm(outPfm.c_str(), cam.width, cam.height, rgb.data()) != MIPT_OK) die("save pfm");
    if (node) mipt_node_destroy(node);
    if (ctx) mipt_context_destroy(ctx);
    mipt_scene_destroy(scene);
    mipt_bvh_destroy(bvh);
    if (env) mipt_image_destroy(env);
    mipt_mesh_destroy(mesh);
    return 0;
}
