// helper.hpp — the parts of the reference's include/helper.hpp that feed the hot path:
// Camera::camera_matrix (defines RenderData.camera_to_world) and MaterialHelper.
#pragma once

#include <string>
#include <vector>

#include "material.hpp"
#include "shape.hpp"

struct Camera {
	glm::vec3 position;
	float yaw;
	float pitch;

	/// translate(position) * eulerAngleYXZ(yaw, pitch, 0): column 3 = camera position
	glm::mat4 camera_matrix() const {
		glm::mat4 camera = glm::translate(glm::mat4(1.0f), position);
		camera *= glm::eulerAngleYXZ(yaw, pitch, 0.0f);
		return camera;
	}
};

struct MaterialHelper {
	std::vector<Material> materials;
	std::vector<std::string> names;

	void remove(int index) {
		materials.erase(materials.begin() + index);
		names.erase(names.begin() + index);
	}
	void push(Material &&material, std::string &&name) {
		materials.push_back(material);
		names.push_back(name);
	}
	int last_index() { return (int)materials.size() - 1; }
	int len() { return (int)materials.size(); }
};

inline float *mptr(glm::mat4 &m) { return glm::value_ptr(m); }
