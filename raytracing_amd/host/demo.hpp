// raytracing_amd/host/demo.hpp -- demo scenes (reference: src/cpp/demo.hpp:17-60 random_scene).
// The rand() draws happen in the order the reference's g++ build makes them (arguments of a constructor call are
// evaluated last-to-first), written out as statements here so the scene does not depend on the compiler.
#pragma once

#include "bvh.hpp"
#include "group.hpp"
#include "material.hpp"
#include "primitive.hpp"
#include "texture.hpp"
#include "transform.hpp"
#include "util.hpp"

#include <vector>

inline group random_scene()
{
    group grp;
    grp.add(new sphere(vec3d(0, -1000, 0), 1000, new lambertian(new checker(vec3d(0, 0, 0), vec3d(1, 1, 1)))));
    for (int i = -11; i < 11; i++) {
        for (int j = -11; j < 11; j++) {
            double choose = random_double();
            double cz = j + 0.9 * random_double();
            double cx = i + 0.9 * random_double();
            vec3d center(cx, 0.2, cz);
            if ((center - vec3d(4, 0.2, 0)).length() <= 0.9) continue;
            if (choose < 0.8) {
                vec3d second = random_vector();
                vec3d first = random_vector();
                material* m = new lambertian(first * second);
                vec3d center2 = center + vec3d(0, random_double(0, 0.5), 0);
                grp.add(new msphere(center, center2, 0, 1, 0.2, m));
            }
            else if (choose < 0.95) {
                vec3d albedo = random_vector(0.5, 1);
                double roughness = random_double(0, 0.5);
                grp.add(new sphere(center, 0.2, new metal(albedo, roughness)));
            }
            else {
                grp.add(new sphere(center, 0.2, new dielectric(1.5)));
            }
        }
    }
    grp.add(new sphere(vec3d(0, 1, 0), 1.0, new dielectric(1.5)));
    grp.add(new sphere(vec3d(-4, 1, 0), 1.0, new lambertian(vec3d(0.4, 0.2, 0.1))));
    grp.add(new sphere(vec3d(4, 1, 0), 1.0, new metal(vec3d(0.7, 0.6, 0.5), 0)));
    return group(new bvh(grp, 0, 1));
}

// BASELINE config 4 ("dielectric/metal-heavy deep-bounce scene"): random_scene's layout with the small spheres' materials
// remapped to 50 % dielectric, 40 % mirror metal (roughness 0), 10 % lambertian.  Not in the reference (its demo.hpp has
// no such scene): defined by SURVEY 8d to stress wave divergence with long specular chains (max_depth 200).
inline group deep_bounce_scene()
{
    group grp;
    grp.add(new sphere(vec3d(0, -1000, 0), 1000, new lambertian(new checker(vec3d(0.1, 0.1, 0.1), vec3d(0.9, 0.9, 0.9)))));
    for (int i = -11; i < 11; i++) {
        for (int j = -11; j < 11; j++) {
            double choose = random_double();
            double cz = j + 0.9 * random_double();
            double cx = i + 0.9 * random_double();
            vec3d center(cx, 0.2, cz);
            if ((center - vec3d(4, 0.2, 0)).length() <= 0.9) continue;
            if (choose < 0.5) grp.add(new sphere(center, 0.2, new dielectric(1.5)));
            else if (choose < 0.9) grp.add(new sphere(center, 0.2, new metal(random_vector(0.5, 1), 0)));
            else grp.add(new sphere(center, 0.2, new lambertian(random_vector() * random_vector())));
        }
    }
    grp.add(new sphere(vec3d(0, 1, 0), 1.0, new dielectric(1.5)));
    grp.add(new sphere(vec3d(-4, 1, 0), 1.0, new metal(vec3d(0.9, 0.9, 0.9), 0)));
    grp.add(new sphere(vec3d(4, 1, 0), 1.0, new metal(vec3d(0.7, 0.6, 0.5), 0)));
    return group(new bvh(grp, 0, 1));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The reference's two remaining demo scenes as DATA: what they hold is the reference's (src/cpp/demo.hpp:63-91 cornell_box, :94-143
// all_feature_test -- the numbers have to be its numbers, the bit-exact tree fixtures tests/golden/cornell.npz and features.npz pin
// them), how they are put together is this file's: a palette, a list of parts, one assembler.  A part that draws random numbers
// (the field of boxes, the cloud of spheres) draws them when the assembler reaches it, so rand() is consumed in list order -- the
// reference's order -- and the trees come out node for node.
namespace demo_parts {

struct paint { char kind; double r, g, b, value; };      // 'L' lambertian | 'M' metal (value: roughness) | 'G' dielectric (value: index) | 'E' emissive
                                                          // | 'I' lambertian over an image file that does not exist (texture.hpp:97-99: cyan)
struct part {
    enum what_t { rect_xy, rect_xz, rect_yz, ball, moving_ball, block, medium_in_ball, ball_with_medium, box_field, ball_cloud } what;
    double n[8];          // rect: a0 a1 b0 b1 k | ball: centre, radius | moving_ball: centre, radius, travel in x | block: size xyz, turn (degrees), place xyz
                          // medium_in_ball / ball_with_medium: centre, radius, density, fog colour | box_field: cells, pitch, origin, lowest and highest top
                          // ball_cloud: count, extent, radius, turn (degrees), place xyz
    int paint;            // palette row (media: the boundary sphere's glass)
    bool flipped;
};

inline material* mix(const paint& p)
{
    switch (p.kind) {
    case 'L': return new lambertian(vec3d(p.r, p.g, p.b));
    case 'M': return new metal(vec3d(p.r, p.g, p.b), p.value);
    case 'G': return new dielectric(p.value);
    case 'E': return new emissive(vec3d(p.r, p.g, p.b));
    default:  return new lambertian(new image("C:/Users/Cronix/Pictures/image.jpg"));
    }
}

// palette rows are mixed once and shared by every part that names them (the reference shares its material pointers the same way)
inline group assemble(const paint* palette, int n_paints, const part* parts, int n_parts)
{
    std::vector<material*> pot(n_paints, nullptr);
    auto colour = [&](int i) { if (!pot[i]) pot[i] = mix(palette[i]); return pot[i]; };
    group out;
    for (int i = 0; i < n_parts; i++) {
        const part& p = parts[i];
        const double* n = p.n;
        object* made = nullptr;
        switch (p.what) {
        case part::rect_xy: made = new planexy(n[0], n[1], n[2], n[3], n[4], colour(p.paint)); break;
        case part::rect_xz: made = new planexz(n[0], n[1], n[2], n[3], n[4], colour(p.paint)); break;
        case part::rect_yz: made = new planeyz(n[0], n[1], n[2], n[3], n[4], colour(p.paint)); break;
        case part::ball: made = new sphere(vec3d(n[0], n[1], n[2]), n[3], colour(p.paint)); break;
        case part::moving_ball: made = new msphere(vec3d(n[0], n[1], n[2]), vec3d(n[0], n[1], n[2]) + vec3d(n[4], 0, 0), 0, 1, n[3], colour(p.paint)); break;
        case part::block:
            made = new translate(new rotatey(new box(vec3d(0, 0, 0), vec3d(n[0], n[1], n[2]), colour(p.paint)), n[3]), vec3d(n[4], n[5], n[6]));
            break;
        case part::medium_in_ball:       // the boundary is not an object of the scene: only the medium inside it is
            made = new volume(new sphere(vec3d(n[0], n[1], n[2]), n[3], mix(palette[p.paint])), n[4], vec3d(n[5], n[6], n[7]));
            break;
        case part::ball_with_medium: {   // a glass ball AND the medium it bounds: two objects over one boundary
            object* shell = new sphere(vec3d(n[0], n[1], n[2]), n[3], mix(palette[p.paint]));
            out.add(shell);
            made = new volume(shell, n[4], vec3d(n[5], n[6], n[7]));
            break;
        }
        case part::box_field: {          // cells x cells boxes standing on y = 0, tops drawn per box, x-major like the reference's loops
            group field;
            const int cells = (int) n[0];
            for (int ix = 0; ix < cells; ix++)
                for (int iz = 0; iz < cells; iz++) {
                    const double x = n[2] + ix * n[1], z = n[2] + iz * n[1];
                    const double top = random_double(n[3], n[4]);
                    field.add(new box(vec3d(x, 0, z), vec3d(x + n[1], top, z + n[1]), colour(p.paint)));
                }
            out.add(field);
            continue;
        }
        case part::ball_cloud: {         // count balls at random places in a cube, under a tree of their own, turned and moved as one
            group cloud;
            for (int k = 0; k < (int) n[0]; k++) cloud.add(new sphere(random_vector(0, n[1]), n[2], colour(p.paint)));
            made = new translate(new rotatey(new bvh(cloud), n[3]), vec3d(n[4], n[5], n[6]));
            break;
        }
        }
        out.add(p.flipped ? new flip(made) : made);
    }
    return out;
}

// ---- the Cornell box of the reference's main()
const paint cornell_palette[] = { { 'L', 0.65, 0.05, 0.05, 0 }, { 'L', 0.12, 0.45, 0.15, 0 }, { 'L', 0.75, 0.75, 0.75, 0 }, { 'E', 30, 30, 30, 0 }, { 'M', 1, 1, 1, 0 } };
const part cornell_parts[] = {
    { part::rect_yz, { 0, 555, 0, 555, 555 }, 1, false },                      // green wall
    { part::rect_yz, { 0, 555, 0, 555, 0 }, 0, false },                        // red wall
    { part::rect_xz, { 0, 555, 0, 555, 0 }, 2, false },                        // floor
    { part::rect_xz, { 0, 555, 0, 555, 555 }, 2, false },                      // ceiling
    { part::rect_xy, { 0, 555, 0, 555, 555 }, 2, false },                      // back wall
    { part::rect_xz, { 213, 343, 227, 332, 554 }, 3, true },                   // the lamp, facing down
    { part::block, { 165, 330, 165, 15, 265, 0, 295 }, 4, false },             // tall mirror block
    { part::block, { 165, 165, 165, -18, 130, 0, 65 }, 2, false },             // short white block
};

// ---- "the next week"'s closing scene as the reference builds it
const paint feature_palette[] = { { 'L', 0.73, 0.73, 0.73, 0 }, { 'L', 0.48, 0.83, 0.53, 0 }, { 'E', 7, 7, 7, 0 }, { 'L', 0.7, 0.3, 0.1, 0 }, { 'G', 0, 0, 0, 1.5 },
                                  { 'M', 0.8, 0.8, 0.9, 1 }, { 'I', 0, 0, 0, 0 }, { 'M', 1, 1, 1, 0 } };
const part feature_parts[] = {
    { part::box_field, { 20, 100, -1000, 1, 101 }, 1, false },                 // 400 ground boxes, tops between 1 and 101
    { part::rect_xz, { 123, 423, 147, 412, 554 }, 2, false },                  // the lamp (not flipped: it shines upwards, as in the reference)
    { part::moving_ball, { 400, 400, 200, 50, 100 }, 3, false },
    { part::ball, { 260, 150, 45, 50 }, 4, false },                            // glass
    { part::ball, { 0, 150, 145, 50 }, 5, false },                             // fuzzy metal
    { part::medium_in_ball, { 0, 0, 0, 5000, 0.0002, 1, 1, 1 }, 4, false },    // haze over everything
    { part::ball_with_medium, { 360, 150, 145, 70, 0.2, 0.2, 0.4, 0.9 }, 4, false },   // glass ball full of blue fog
    { part::ball, { 400, 200, 400, 100 }, 6, false },                          // the ball with the picture that is not there
    { part::ball, { 220, 280, 300, 50 }, 7, false },                           // mirror
    { part::ball_cloud, { 1000, 165, 10, 15, -100, 270, 395 }, 0, false },
};

}  // namespace demo_parts

// reference: src/cpp/demo.hpp:94-143 (the image path of :132 exists nowhere: the texture fails to load and is cyan, texture.hpp:97-99)
inline group all_feature_test()
{
    group grp = demo_parts::assemble(demo_parts::feature_palette, 8, demo_parts::feature_parts, 10);
    return group(new bvh(grp, 0, 1));
}

// the sphere-bounded media of all_feature_test() (reference: src/cpp/demo.hpp:117-134: haze in a radius-5000 sphere, a glass ball
// filled with blue fog, moving / glass / fuzzy-metal / mirror spheres) over a ground rect, plus a free-standing smoke ball
inline group smoke_scene()
{
    group grp;
    grp.add(new planexz(-1000, 1000, -1000, 1000, 0, new lambertian(vec3d(0.48, 0.83, 0.53))));
    grp.add(new flip(new planexz(123, 423, 147, 412, 554, new emissive(vec3d(7, 7, 7)))));
    grp.add(new msphere(vec3d(400, 400, 200), vec3d(430, 400, 200), 0, 1, 50, new lambertian(vec3d(0.7, 0.3, 0.1))));
    grp.add(new sphere(vec3d(260, 150, 45), 50, new dielectric(1.5)));
    grp.add(new sphere(vec3d(0, 150, 145), 50, new metal(vec3d(0.8, 0.8, 0.9), 1)));
    grp.add(new volume(new sphere(vec3d(0, 0, 0), 5000, new dielectric(1.5)), 0.0002, vec3d(1, 1, 1)));
    object* ball = new sphere(vec3d(360, 150, 145), 70, new dielectric(1.5));
    grp.add(ball);
    grp.add(new volume(ball, 0.2, vec3d(0.2, 0.4, 0.9)));
    grp.add(new sphere(vec3d(400, 200, 400), 100, new lambertian(new checker(vec3d(0.2, 0.3, 0.1), vec3d(0.9, 0.9, 0.9)))));
    grp.add(new sphere(vec3d(220, 280, 300), 50, new metal(vec3d(1, 1, 1), 0)));
    grp.add(new volume(new sphere(vec3d(150, 100, 250), 80, new dielectric(1.5)), 0.01, vec3d(0.9, 0.9, 0.9)));
    return grp;
}

// the scene of the reference's main() (reference: src/cpp/demo.hpp:63-91)
inline group cornell_box()
{
    return demo_parts::assemble(demo_parts::cornell_palette, 5, demo_parts::cornell_parts, 8);
}

// What the object model composes and the reference's demos never do (no counterpart in demo.hpp; the same scene as oracle/ref_harness.cpp
// build_nested, which builds it from the reference's own classes): wrappers nested every way (transform.hpp:9-149), volumes around boxes
// (primitive.hpp:332-343), groups used as objects (group.hpp:12,68-102), an instance inside an instance.  Leaves in this order are the
// fixture's primitive ids.
inline group nested_test()
{
    group grp;
    grp.add(new planexz(-40, 40, -40, 40, 0, new lambertian(vec3d(0.5, 0.5, 0.5))));
    grp.add(new flip(new planexz(-4, 4, -4, 4, 14, new emissive(vec3d(8, 8, 8)))));
    grp.add(new translate(new translate(new sphere(vec3d(0, 1, 0), 1, new lambertian(vec3d(0.8, 0.3, 0.3))), vec3d(-6, 0, 0)), vec3d(0, 0, -3)));
    grp.add(new translate(new rotatey(new rotatey(new box(vec3d(-1, 0, -1), vec3d(1, 2, 1), new lambertian(vec3d(0.3, 0.8, 0.3))), 20), 25), vec3d(-2, 0, 4)));
    grp.add(new translate(new rotatey(new translate(new rotatey(new sphere(vec3d(0, 1, 0), 1, new dielectric(1.5)), 30), vec3d(3, 0, 0)), 40), vec3d(2, 0, -6)));
    grp.add(new rotatey(new translate(new sphere(vec3d(0, 1.2, 0), 1.2, new metal(vec3d(0.8, 0.8, 0.6), 0.1)), vec3d(7, 0, 0)), 50));
    grp.add(new volume(new box(vec3d(-9, 0, 2), vec3d(-6, 3, 5), new dielectric(1.5)), 0.4, vec3d(0.9, 0.9, 0.9)));
    grp.add(new volume(new translate(new rotatey(new box(vec3d(0, 0, 0), vec3d(2.5, 3, 2.5), new dielectric(1.5)), -18), vec3d(5, 0, 3)), 0.5, vec3d(0.2, 0.4, 0.9)));
    grp.add(new translate(new volume(new sphere(vec3d(0, 1.5, 0), 1.5, new dielectric(1.5)), 0.6, vec3d(0.9, 0.5, 0.2)), vec3d(-3, 0, -8)));
    {
        group* sub = new group();
        sub->add(new sphere(vec3d(-10, 1, -6), 1, new lambertian(vec3d(0.7, 0.7, 0.2))));
        sub->add(new triangle(vec3d(-12, 0, -4), vec3d(-9, 0, -4), vec3d(-10.5, 3, -4.5), new lambertian(vec3d(0.2, 0.7, 0.7))));
        sub->add(new box(vec3d(-13, 0, -9), vec3d(-11, 2, -7), new lambertian(vec3d(0.7, 0.2, 0.7))));
        grp.add(sub);
    }
    {
        group* sub = new group();
        sub->add(new sphere(vec3d(0, 1, 0), 1, new lambertian(vec3d(0.9, 0.6, 0.3))));
        sub->add(new sphere(vec3d(2.2, 1, 0), 1, new metal(vec3d(0.9, 0.9, 0.9), 0.0)));
        grp.add(new translate(sub, vec3d(8, 0, -9)));
    }
    {
        group in;
        in.add(new sphere(vec3d(0, 0.6, 0), 0.6, new lambertian(vec3d(0.9, 0.2, 0.2))));
        in.add(new sphere(vec3d(1.5, 0.6, 0), 0.6, new lambertian(vec3d(0.2, 0.9, 0.2))));
        in.add(new sphere(vec3d(0, 0.6, 1.5), 0.6, new lambertian(vec3d(0.2, 0.2, 0.9))));
        object* inner = new bvh(in);
        group mid;
        mid.add(new sphere(vec3d(-2, 1, 0), 1, new lambertian(vec3d(0.6, 0.6, 0.6))));
        mid.add(new translate(inner, vec3d(0, 2.2, 0)));
        mid.add(new sphere(vec3d(2.5, 1, 1), 1, new dielectric(1.5)));
        grp.add(new translate(new rotatey(new bvh(mid), 35), vec3d(-9, 0, 9)));
    }
    grp.add(new flip(new translate(new sphere(vec3d(0, 1, 0), 1, new dielectric(1.5)), vec3d(4, 0, 8))));
    grp.add(new translate(new flip(new sphere(vec3d(0, 1, 0), 1, new dielectric(1.5))), vec3d(7, 0, 7)));
    {
        group m;
        m.add(new sphere(vec3d(11, 1, 2), 1, new dielectric(1.5)));
        m.add(new planexy(9, 13, 0, 3, 0, new lambertian(vec3d(0.4, 0.4, 0.8))));
        grp.add(new flip(new bvh(m)));
    }
    grp.add(new translate(new msphere(vec3d(0, 1, 0), vec3d(0, 1.5, 0), 0, 1, 0.8, new lambertian(vec3d(0.7, 0.3, 0.1))), vec3d(-5, 0, 6)));
    return grp;
}
