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

// reference: src/cpp/demo.hpp:94-143, statement for statement (the image path of :132 does not exist anywhere: the texture
// fails to load and is cyan, texture.hpp:97-99)
inline group all_feature_test()
{
    group grp;
    group ground_boxes;
    material* white_mat = new lambertian(vec3d(0.73, 0.73, 0.73));
    material* ground_mat = new lambertian(vec3d(0.48, 0.83, 0.53));
    material* area_light = new emissive(vec3d(7, 7, 7));
    for (int i = 0; i < 20; i++) {
        for (int j = 0; j < 20; j++) {
            double w = 100;
            double x0 = -1000 + i * w;
            double z0 = -1000 + j * w;
            double y0 = 0;
            double x1 = x0 + w;
            double z1 = z0 + w;
            double y1 = random_double(1, 101);
            ground_boxes.add(new box(vec3d(x0, y0, z0), vec3d(x1, y1, z1), ground_mat));
        }
    }
    grp.add(ground_boxes);
    grp.add(new planexz(123, 423, 147, 412, 554, area_light));
    vec3d center_start = vec3d(400, 400, 200);
    vec3d center_end = center_start + vec3d(100, 0, 0);
    material* motion_ball_mat = new lambertian(vec3d(0.7, 0.3, 0.1));
    grp.add(new msphere(center_start, center_end, 0, 1, 50, motion_ball_mat));
    grp.add(new sphere(vec3d(260, 150, 45), 50, new dielectric(1.5)));
    grp.add(new sphere(vec3d(0, 150, 145), 50, new metal(vec3d(0.8, 0.8, 0.9), 1)));
    object* background = new sphere(vec3d(0, 0, 0), 5000, new dielectric(1.5));
    grp.add(new volume(background, 0.0002, vec3d(1, 1, 1)));
    background = new sphere(vec3d(360, 150, 145), 70, new dielectric(1.5));
    grp.add(background);
    grp.add(new volume(background, 0.2, vec3d(0.2, 0.4, 0.9)));
    material* img_mat = new lambertian(new image("C:/Users/Cronix/Pictures/image.jpg"));
    grp.add(new sphere(vec3d(400, 200, 400), 100, img_mat));
    grp.add(new sphere(vec3d(220, 280, 300), 50, new metal(vec3d(1, 1, 1), 0)));
    group vol_boxes;
    for (int i = 0; i < 1000; i++) {
        vol_boxes.add(new sphere(random_vector(0, 165), 10, white_mat));
    }
    grp.add(new translate(new rotatey(new bvh(vol_boxes), 15), vec3d(-100, 270, 395)));
    bvh* bvh_grp = new bvh(grp, 0, 1);
    return group(bvh_grp);
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
    group grp;
    material* red_diffuse = new lambertian(vec3d(0.65, 0.05, 0.05));
    material* green_diffuse = new lambertian(vec3d(0.12, 0.45, 0.15));
    material* white_diffuse = new lambertian(vec3d(0.75, 0.75, 0.75));
    material* area_light = new emissive(vec3d(30, 30, 30));
    material* mirror = new metal(vec3d(1, 1, 1), 0);
    grp.add(new planeyz(0, 555, 0, 555, 555, green_diffuse));
    grp.add(new planeyz(0, 555, 0, 555, 0, red_diffuse));
    grp.add(new planexz(0, 555, 0, 555, 0, white_diffuse));
    grp.add(new planexz(0, 555, 0, 555, 555, white_diffuse));
    grp.add(new planexy(0, 555, 0, 555, 555, white_diffuse));
    grp.add(new flip(new planexz(213, 343, 227, 332, 554, area_light)));
    object* box1 = new box(vec3d(0, 0, 0), vec3d(165, 330, 165), mirror);
    box1 = new rotatey(box1, 15);
    box1 = new translate(box1, vec3d(265, 0, 295));
    grp.add(box1);
    object* box2 = new box(vec3d(0, 0, 0), vec3d(165, 165, 165), white_diffuse);
    box2 = new rotatey(box2, -18);
    box2 = new translate(box2, vec3d(130, 0, 65));
    grp.add(box2);
    return grp;
}
