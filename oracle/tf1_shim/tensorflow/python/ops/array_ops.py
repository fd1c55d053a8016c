"""tf1_shim: placeholder for `from tensorflow.python.ops import array_ops`."""
