"""tf1_shim: placeholder for `from tensorflow.python.framework import dtypes`."""
