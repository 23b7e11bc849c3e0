(( Leiden University (c) 2002
 ---- Problem ---- 
<	i1	i2	i3	i4	i5	i6	i7	i8	i9	i10><><	C><	j1	j2	BigPar>
 ---- Context ---- 
<	j1	j2	BigPar><	C>
