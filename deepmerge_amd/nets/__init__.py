"""Mirror of the reference's `nets` package (one module: ShfitScaleFormer)."""
