/* placeholder translation unit; LogNormal restatement lands here (statistical_models.py:998-1160, minimizer.py) */
int fo_lognormal_placeholder(void) { return 0; }
