// raytracing_amd/host/group.hpp -- object list (reference: src/cpp/group.hpp:12-124).
// Copying a group copies the pointer list only (shallow), as in the reference.
#pragma once

#include <vector>

#include "object.hpp"

class group : public object {
public:
    group() {}
    group(object* obj) { add(obj); }
    void add(object* obj) { items.push_back(obj); }
    void add(const group& other) { for (object* o : other.items) items.push_back(o); }
    void clear() { items.clear(); }
    int size() const { return (int) items.size(); }
    object** content() { return items.data(); }
    object* const* content() const { return items.data(); }
    // union with a zero box first, as the reference's `first = false` does (group.hpp:91-99)
    bool bounding_box(double start, double end, aabb& bbox) const override
    {
        if (items.empty()) return false;
        aabb tmp;
        for (object* o : items) {
            if (!o->bounding_box(start, end, tmp)) return false;
            bbox = surrounding_box(bbox, tmp);
        }
        return true;
    }
private:
    std::vector<object*> items;
};
