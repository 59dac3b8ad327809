//! HIP back end for the path-tracing step: a drop-in for the headless subset of `crate::vulkan` (src/vulkan/mod.rs:1-15).
//! No HIP or C types escape this module, as no vulkanalia types escape `vulkan`.
pub mod ffi;
pub mod renderer;

pub use renderer::HipRenderer;
