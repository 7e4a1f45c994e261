"""oak_amd: MI355X-native batched RBY rollout + leaf-evaluation engine (drop-in for the
random-playout / leaf-eval hot path of lab-oak/oak).  See DESIGN.md."""
__version__ = "0.1.0"
