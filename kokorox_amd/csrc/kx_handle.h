// The opaque C handle shared by api.hip and dispatcher.hip.
#pragma once
#include <memory>

#include "model.h"

struct kx_model {
    std::unique_ptr<kx::Model> m;
};
