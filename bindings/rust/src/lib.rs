//! `rttnw-hip`: what `src/math/` of luliic2/rttnw becomes when the tracing core runs on an MI355X.
//!
//! * [`ffi`] — `include/rttnw_hip.h`, declaration for declaration (held against the header by a test).
//! * [`scene`] — handles and a builder with the vocabulary of the reference's `scenes.rs`
//!   (`Sphere {..}` -> `b.sphere(..)`, `.rotate_y(a).translate(v)` -> `b.translate(b.rotate_y(x, a), v)`), and
//!   `render()` as one call.
pub mod ffi;
pub mod scene;
