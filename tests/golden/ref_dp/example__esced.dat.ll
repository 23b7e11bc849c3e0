(( Programme esced :
   Code :
        do i=1,n
   (S1)    a(i) = i
           do j=1,m
   (S2)       b(j) = b(j) + a(i)
           endo
        endo
    
   Farkas :
   O(S1,i)   = n1 + n2(i-1) + n3(n-i)
   O(S2,i,j) = n4 + n5(i-1) + n6(n-i) + n7(j-1) + n8(m-j)

   (-1)*n1 +  (1)*n4 >= 0
   (-1)*n2 +  (1)*n5 >= 0
   (-1)*n3 +  (1)*n5 >= 0 
    (1)*n5 + (-1)*n6 >= 0
              (1)*n7 >= 0
              (1)*n8 >= 0
    
   Chunking :
   TS1 = [ [ 1 ] ]
   TS2 = [ [ 0 1 ] [ 0 0 ]
    
   Decalages :
   - pour TS1
    (1)*n1 + (-1)*n2 == b1
              (1)*n3 == bn1
   - pour TS2
    (1)*n4 + (-1)*n5 + (-1)*n7 == b2
              (1)*n6 == bn2
              (1)*n8 == bm2
    
  Construction :
   - pour TS1
    (1)*n2 + (-1)*n3 == (1)*CS1,1
   - pour TS2
    (1)*n5 + (-1)*n6 == (0)*CS2,1 + (0)*CS2,2
    (1)*n7 + (-1)*n8 == (1)*CS2,1 + (0)*CS2,2
     
  Non nullite (simplifiee) :
                (1)*CS1,1 >= 1
    (1)*CS2,1 + (1)*CS2,2 >= 1
 )
(list #[ 1]
#[ 0]
#[ 1]
#[ 0]
#[ 0]
#[ 0]
#[ 1]
#[ 0]
#[ 1]
#[ 1]
#[ 0]
#[ 1]
#[ 1]
#[ 1]
#[ 0]
#[ 0]
)
)
