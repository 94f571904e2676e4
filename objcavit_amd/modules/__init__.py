"""Drop-in counterparts of the reference's ``modules`` package for the forward
depth-inference hot path (same module file names and class names)."""
