"""Mirror of the reference's `lib` package (module and attribute names, state_dict keys) on the HIP engine."""
